#!/bin/bash
OUT=gpurun_out/quad_sweep.txt
: > $OUT
for rep in 1 2 3; do
  for cfg in "2 2 3" "4 4 4" "4 4 5" "4 4 6" "4 3 4" "4 5 6" "4 6 7"; do
    set -- $cfg
    v=$(SLAM_ICP_TEAMS=$1 timeout -k 10 120 python3 bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline --step-streams $2 --grid-lag $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms']['raycast']))")
    echo "teams $1 streams $2 lag $3 rep $rep: $v" >> $OUT
  done
done
sort $OUT
