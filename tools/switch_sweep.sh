#!/bin/bash
# iterations run by the ring-search launch before the list-sweep launch takes over
#   usage: bash tools/switch_sweep.sh [scans_per_gpu]
S=${1:-256}
for K in 4 5 6 8 10 12 16; do
  SLAM_ICP_SWITCH_ITER=$K timeout -k 10 120 python bench.py --no-torch --steps 10 --warmup 2 --no-cpu-baseline --scans $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scans=$S switch_iter=$K', {k: round(v,4) for k,v in d['kernel_ms'].items()}, round(d['value']/1e6,1))" || exit 1
done
