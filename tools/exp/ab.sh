#!/bin/bash
# A/B of two builds of the library on the GPU box: tools/exp/libslam_base.so (baseline) against slam_amd/lib/libslam_mi355x.so
OUT=gpurun_out/ab.txt
: > $OUT
cp slam_amd/lib/libslam_mi355x.so /tmp/new.so
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then cp tools/exp/libslam_base.so slam_amd/lib/libslam_mi355x.so; else cp /tmp/new.so slam_amd/lib/libslam_mi355x.so; fi
    echo "== $which rep $rep" >> $OUT
    timeout -k 10 200 python3 tools/pair_time.py 256 512 1024 2>/dev/null | grep "pair= 2\|pair=-1" >> $OUT
  done
done
cp /tmp/new.so slam_amd/lib/libslam_mi355x.so
cat $OUT
