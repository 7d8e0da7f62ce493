#!/bin/bash
# A/B of two builds on the pipelined step: tools/exp/libslam_base.so against slam_amd/lib/libslam_mi355x.so
OUT=gpurun_out/ab_step.txt
: > $OUT
cp slam_amd/lib/libslam_mi355x.so /tmp/new.so
for rep in 1 2 3 4; do
  for which in base new; do
    if [ $which = base ]; then cp tools/exp/libslam_base.so slam_amd/lib/libslam_mi355x.so; else cp /tmp/new.so slam_amd/lib/libslam_mi355x.so; fi
    v=$(timeout -k 10 120 python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    w=$(timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % (d['ms_per_step']))")
    echo "$which rep $rep K50 $v K20 $w" >> $OUT
  done
done
cp /tmp/new.so slam_amd/lib/libslam_mi355x.so
sort $OUT
