#!/bin/bash
# config 3's C++ sequence program by itself: the three forms, and a kernel trace of the batch form (round 5)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O/data
python tools/exp/c3_data.py $O/data 50 10
g++ -std=c++17 -O2 -pthread -I include tests/cpp/ccicp_sequence.cpp -o $O/ccicp_sequence -L slam_amd/lib -l:libslam_mi355x.so -Wl,-rpath,$PWD/slam_amd/lib -Wl,-rpath,/opt/rocm/lib
for f in seq ahead batch; do $O/ccicp_sequence $O/data 50 10 3 $f; done | tee $O/forms.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -- $O/ccicp_sequence $O/data 50 10 2 batch > $O/trace_batch.txt 2>&1
python - <<P
import csv, glob
f = sorted(glob.glob("$O/trace_batch/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r["Queue_Id"]) for r in rows)
# the last batch: from the last pack_scans_kernel back to the chains before it
packs = [i for i, k in enumerate(K) if "pack_scans" in k[2]]
i1 = packs[-2]; i0 = packs[-3]
t0 = K[i0][1]
print("between two packs: %.1f us, %d kernels" % ((K[i1][0] - t0) / 1e3, i1 - i0))
busy = sum(k[1] - k[0] for k in K[i0:i1])
print("sum of kernel durations %.1f us" % (busy / 1e3))
for k in K[i0:i1 + 3]:
    if (k[1] - k[0]) > 20000 or "pack" in k[2] or "spread" in k[2] or "height_fit" in k[2]:
        print("%9.1f us %8.1f us q%s %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
P
