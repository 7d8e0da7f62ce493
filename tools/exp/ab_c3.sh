#!/bin/bash
# A/B of two builds of the library on config 3's C++ adapter forms: tools/exp/libslam_base.so (baseline) against slam_amd/lib/libslam_mi355x.so
OUT=gpurun_out/ab_c3.txt
: > $OUT
cp slam_amd/lib/libslam_mi355x.so /tmp/new.so
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then cp tools/exp/libslam_base.so slam_amd/lib/libslam_mi355x.so; else cp /tmp/new.so slam_amd/lib/libslam_mi355x.so; fi
    timeout -k 10 200 python3 bench.py --config 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cpp_adapter']
print('$which rep $rep seq', c['ms_per_match'], {k:v['ms_per_match'] for k,v in c['throughput_forms'].items() if isinstance(v,dict)})" >> $OUT
  done
done
cp /tmp/new.so slam_amd/lib/libslam_mi355x.so
cat $OUT
