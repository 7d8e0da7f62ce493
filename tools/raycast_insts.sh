#!/bin/bash
# VALU / SALU / LDS instruction counts of raycast_tiled_kernel with parts of it switched off (measurement build:
# SLAM_RAYCAST_ABLATE bits 1 = walk, 2 = write-back, 8 = no blocks at all, 16 = clipping): where the instructions go
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SLAM_AMD_MEASURE=1
for A in 0 1 2 3 8 16; do
  export SLAM_RAYCAST_ABLATE=$A; rm -rf gpurun_out/rinst_$A
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/rinst_$A -- python3 tools/raycast_time.py > gpurun_out/rinst_$A.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/rinst_$A/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "raycast_tiled" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate=$A", {k: "%.3g" % (sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
done
