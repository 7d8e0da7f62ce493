#!/bin/bash
# registration launch alone on the chip (one stream) for the large models, tiled / untiled (SLAM_BENCH_TILES=-1 in bench's env -> wave_tiles)
for m in "20000 room" "39998 room" "39998 uniform"; do set -- $m
  for t in 1 0; do
    echo -n "$1 $2 wave_tiles=$t: "
    timeout 100 python bench.py --map-points $1 --map-kind $2 --wave-tiles $t --no-pipeline --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"kernel_ms\"][\"icp\"],4), d[\"max_pose_error_m\"])"
  done
done
