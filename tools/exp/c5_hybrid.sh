#!/bin/bash
# config 5: pairs on two streams except the N chunks from a rebuild's begin on, which register one scan per workgroup (SLAM_MAPPER_HYBRID=N)
OUT=gpurun_out/c5_hybrid.txt; : > $OUT
run() {
  v=$(env $2 timeout -k 10 200 python3 bench.py --config 5 --stream-scans 10240 $3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['config'].get('mapper',{})
print('%.4f ms per chunk  %.1f M points/s  rebuilds %s rebuild_ms %.2f  err %.4f' % (d['ms_per_step'], d['value']/1e6, m.get('rebuilds'), m.get('rebuild_ms',0), d['max_pose_error_m']))")
  echo "$1: $v" >> $OUT
}
for rep in 1 2 3; do
  run "default (one stream, one scan per workgroup)" "X=1" ""
  run "pairs on two streams" "X=1" "--reg-streams 2 --pair-scans 2"
  run "hybrid 1" "SLAM_MAPPER_HYBRID=1" "--reg-streams 2 --pair-scans 2"
  run "hybrid 2" "SLAM_MAPPER_HYBRID=2" "--reg-streams 2 --pair-scans 2"
  run "hybrid 3" "SLAM_MAPPER_HYBRID=3" "--reg-streams 2 --pair-scans 2"
done
cat $OUT
