#!/bin/bash
# iteration from which a scan may leave the ring search for the list sweeps, and the share of far queries that
# still allows it
#   usage: bash tools/switch_sweep.sh [scans_per_gpu]
S=${1:-256}
for D in 32 128; do
for K in 3 4 5 6 8 10 12; do
  SLAM_ICP_FAR_DIV=$D SLAM_ICP_SWITCH_ITER=$K timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --scans $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scans=$S far_div=$D switch_iter=$K', {k: round(v,4) for k,v in d['kernel_ms'].items()}, round(d['value']/1e6,1))" || exit 1
done
done
