#!/bin/bash
# config 5 in pairs on two registration streams against the rebuild cadence: is it the REBUILDS (their launches beside pair workgroups) or the
# REBUILT MODEL (prior + window, 10 k points) that separates it from the never-rebuilt 0.306 ms per chunk?
OUT=gpurun_out/c5_cadence.txt; : > $OUT
run() {
  v=$(timeout -k 10 200 python3 bench.py --config 5 --stream-scans 10240 $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['config'].get('mapper',{})
print('%.4f ms per chunk  %.1f M points/s  rebuilds %s rebuild_ms %.2f' % (d['ms_per_step'], d['value']/1e6, m.get('rebuilds'), m.get('rebuild_ms',0)))")
  echo "$1: $v" >> $OUT
}
for rep in 1 2; do
  run "default (one stream, one scan per workgroup, rebuild every 4)" ""
  run "one stream, rebuild every 16" "--rebuild-every 16"
  run "one stream, never rebuilt" "--rebuild-every 100000"
  run "pairs on two streams, rebuild every 4" "--reg-streams 2 --pair-scans 2"
  run "pairs on two streams, rebuild every 16" "--reg-streams 2 --pair-scans 2 --rebuild-every 16"
  run "pairs on two streams, rebuild every 48" "--reg-streams 2 --pair-scans 2 --rebuild-every 48"
  run "pairs on two streams, never rebuilt" "--reg-streams 2 --pair-scans 2 --rebuild-every 100000"
done
cat $OUT
