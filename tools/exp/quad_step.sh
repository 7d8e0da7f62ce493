#!/bin/bash
# measurement: four scans per workgroup (SLAM_TEAMS=4) in the pipelined step, with 2/3/4 registration streams, against pairs
OUT=gpurun_out/quad_step.txt
: > $OUT
run() { # label, env, args
  v=$(env $2 timeout -k 10 150 python3 bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  launch %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "$1: $v" >> $OUT
}
for rep in 1 2; do
  run "pairs  2 streams" "X=1" "--step-streams 2"
  run "pairs  3 streams" "X=1" "--step-streams 3"
  run "quads  2 streams" "SLAM_TEAMS=4" "--step-streams 2"
  run "quads  3 streams" "SLAM_TEAMS=4" "--step-streams 3"
  run "quads  4 streams" "SLAM_TEAMS=4" "--step-streams 4"
  run "quads  5 streams" "SLAM_TEAMS=4" "--step-streams 5"
done
cat $OUT
