#!/bin/bash
# the raycast walk with its lanes staggered: libraries built with -DSLAM_WALK_STAGGER=.. into tools/exp/stag/lib_*.so against each other (round 5)
cd $GRAFT_REPO_ROOT
cp slam_amd/lib/libslam_mi355x.so /tmp/keep.so
for rep in 1 2; do for f in tools/exp/stag/lib_*.so; do
  cp $f slam_amd/lib/libslam_mi355x.so
  echo -n "$(basename $f .so | sed 's/lib_/stagger_steps /'): "
  timeout -k 10 100 python tools/raycast_time.py wg_per_cu=1,2 2>/dev/null | awk '{printf "%s %s ms | ", $1, $6}'
  timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('step %.4f ms raycast in kernel_ms %.4f' % (d['ms_per_step'], d['kernel_ms']['raycast']))"
done; done
cp /tmp/keep.so slam_amd/lib/libslam_mi355x.so
