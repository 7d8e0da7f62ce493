#!/bin/bash
OUT=gpurun_out/streams_sweep.txt
: > $OUT
for rep in 1 2 3; do
  for cfg in "2 3" "3 3" "3 4" "4 4" "4 5" "3 3 --private-queues" "4 4 --private-queues"; do
    set -- $cfg
    v=$(timeout -k 10 120 python3 bench.py --steps 60 --warmup 6 --no-extras --no-cpu-baseline --step-streams $1 --grid-lag $2 $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms']['raycast']))")
    echo "streams $1 lag $2 $3 rep $rep: $v" >> $OUT
  done
done
sort $OUT
