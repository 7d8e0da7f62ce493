#!/bin/bash
# the list form's window (entries either side of the start: 3 / 4 / 5 = shipped / 6 / 7) and walk (2 instead of 3), library builds side by side
OUT=gpurun_out/win_ab.txt
: > $OUT
cp slam_amd/lib/libslam_mi355x.so /tmp/new.so
for rep in 1 2; do
  for V in new w3 w4 w6 w7 k2; do
    if [ $V = new ]; then cp /tmp/new.so slam_amd/lib/libslam_mi355x.so; else cp tools/exp/libslam_$V.so slam_amd/lib/libslam_mi355x.so; fi
    echo "$V rep $rep: $(timeout -k 10 200 python3 tools/exp/win_time.py 2>/dev/null | tail -1)" >> $OUT
  done
done
cp /tmp/new.so slam_amd/lib/libslam_mi355x.so
sort $OUT
