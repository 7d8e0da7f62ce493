#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in 1 33 65 97; do
  export SLAM_RAYCAST_ABLATE=$A; rm -rf gpurun_out/abl_$A
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$A -- python bench.py --no-torch --steps 20 --warmup 3 --no-cpu-baseline --lanes 4 > gpurun_out/abl_$A.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl_$A/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "raycast_tiled" in r["Name"] or "tile_items" in r["Name"]: print("A=$A", r["Name"][22:60].ljust(40), r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
