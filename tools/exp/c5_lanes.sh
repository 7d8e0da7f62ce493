#!/bin/bash
OUT=gpurun_out/c5_lanes.txt
: > $OUT
for rep in 1 2; do
for re in 3 5 8; do
  for sl in 4 5 6; do
    v=$(timeout -k 10 200 python3 bench.py --config 5 --reg-streams 1 --slots $sl --rebuild-every $re 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['ms_per_step'], d['max_pose_error_m']))")
    echo "rebuild_every $re slots $sl rep $rep ms_per_chunk $v" >> $OUT
  done
done
done
sort $OUT
