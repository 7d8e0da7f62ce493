#!/bin/bash
# the point-to-line step with the hand-over after 4 first iterations (new default) against 10 (slam_icp_params via bench.py has no flag: base library)
OUT=gpurun_out/p2l_switch.txt
: > $OUT
cp slam_amd/lib/libslam_mi355x.so /tmp/new.so
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then cp tools/exp/libslam_base.so slam_amd/lib/libslam_mi355x.so; else cp /tmp/new.so slam_amd/lib/libslam_mi355x.so; fi
    v=$(timeout -k 10 120 python3 bench.py --mode p2l --steps 50 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    echo "$which rep $rep p2l step, launch: $v" >> $OUT
  done
done
cp /tmp/new.so slam_amd/lib/libslam_mi355x.so
sort $OUT
