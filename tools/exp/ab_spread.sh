#!/bin/bash
# A/B of two builds of the library on the one-scan fits (tools/spread_time.py): tools/exp/libslam_base.so against slam_amd/lib/libslam_mi355x.so
OUT=gpurun_out/ab_spread.txt; : > $OUT
cp slam_amd/lib/libslam_mi355x.so /tmp/new.so
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then cp tools/exp/libslam_base.so slam_amd/lib/libslam_mi355x.so; else cp /tmp/new.so slam_amd/lib/libslam_mi355x.so; fi
    timeout -k 10 200 python3 tools/spread_time.py 100 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1])
print('$which rep $rep', [d['config3_cloud%d'%k]['us_per_fit_median'] for k in (1,5,9)], [x['us_per_fit_median'] for x in d['room_2x19999']['scans']], [x['us_per_fit_median'] for x in d['room_10k']['scans']])" >> $OUT
  done
done
cp /tmp/new.so slam_amd/lib/libslam_mi355x.so
cat $OUT
