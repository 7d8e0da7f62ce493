#!/bin/bash
OUT=gpurun_out/c5_lanes.txt
: > $OUT
for rep in 1 2; do
  for cfg in "-1 2 5" "-1 2 6" "-1 2 4" "-1 1 5" "0 1 5"; do
    set -- $cfg
    v=$(C5_PAIR=$1 timeout -k 10 200 python3 bench.py --config 5 --reg-streams $2 --slots $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['ms_per_step'], d['max_pose_error_m']))")
    echo "pair $1 streams $2 slots $3 rep $rep ms_per_chunk $v" >> $OUT
  done
done
sort $OUT
