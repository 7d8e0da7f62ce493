#!/bin/bash
run() { timeout -k 10 120 python bench.py --no-torch --steps 10 --warmup 2 --no-cpu-baseline --lanes 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
for A in 0 1 9; do SLAM_RAYCAST_ABLATE=$A run "ablate=$A"; done
for S in 16 32 128; do SLAM_RAYCAST_SEG=$S run "seg=$S"; done
SLAM_RAYCAST_WGPCU=1 run "wgpcu=1"
