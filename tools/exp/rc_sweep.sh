#!/bin/bash
# the pipelined step against the raycast's workgroup count, both solvers (the grid update costs the point-to-line step three times
# what it costs the point-to-point step: tools/exp/two_lane_time.py)
OUT=gpurun_out/rc_sweep.txt
: > $OUT
for rep in 1 2 3; do
  for mode in p2p p2l; do
    for cfg in "0 0" "1 0" "0 256" "0 128" "0 64" "1 128"; do
      set -- $cfg
      v=$(timeout -k 10 120 python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --mode $mode --raycast-wg $1 --raycast-max-wg $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms']['raycast']))")
      echo "$mode wg_per_cu $1 max_wg $2 rep $rep: $v" >> $OUT
    done
  done
done
sort $OUT
