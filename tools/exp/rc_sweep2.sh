#!/bin/bash
OUT=gpurun_out/rc_sweep2.txt
: > $OUT
for rep in 1 2 3; do
  for wg in 0 1; do
    v=$(timeout -k 10 200 python3 bench.py --config 5 --raycast-wg $wg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
    w=$(timeout -k 10 200 python3 bench.py --config 4 --steps 12 --warmup 3 --no-extras --no-cpu-baseline --raycast-wg $wg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['ms_per_step'], d['kernel_ms']['raycast']))")
    x=$(timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --raycast-wg $wg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f %.1f' % (d['ms_per_step'], d['pcie_inclusive']['ms_per_chunk'], d['value_pcie_inclusive']/1e6))")
    echo "wg_per_cu $wg rep $rep: config5 $v | config4 $w | K20 step, pcie chunk, Mpts $x" >> $OUT
  done
done
sort $OUT
