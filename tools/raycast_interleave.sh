#!/bin/bash
# The write-back of the tiled raycast with hits and misses in two planes (as built) against {misses, hits} interleaved per cell
# (one 8-byte atomic per touched cell: measurement build, SLAM_RAYCAST_ABLATE=128, counts wrong): atomic requests to memory and
# bytes written per launch, and the time.   bash tools/raycast_interleave.sh   (needs python -m slam_amd.build --measure)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SLAM_AMD_MEASURE=1
for A in 0 128; do
  export SLAM_RAYCAST_ABLATE=$A
  for PMC in "TCC_EA0_ATOMIC_sum WRITE_SIZE" "FETCH_SIZE"; do
    rm -rf gpurun_out/ileave_$A
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d gpurun_out/ileave_$A -- python3 tools/raycast_time.py > gpurun_out/ileave_$A.log 2>&1
    python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/ileave_$A/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "raycast_tiled" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate=$A", {k: "%.4g" % (sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
  done
  unset -v X; python3 tools/raycast_time.py | tail -1
done
