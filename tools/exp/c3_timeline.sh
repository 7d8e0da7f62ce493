#!/bin/bash
# one batch of CCICP::matchSequence on the device: copies and kernels per queue (round 5)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O/data
python tools/exp/c3_data.py $O/data 50 10
g++ -std=c++17 -O2 -pthread -I include tests/cpp/ccicp_sequence.cpp -o $O/ccicp_sequence -L slam_amd/lib -l:libslam_mi355x.so -Wl,-rpath,$PWD/slam_amd/lib -Wl,-rpath,/opt/rocm/lib
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- $O/ccicp_sequence $O/data 50 10 3 batch > $O/tl.txt 2>&1
python - <<P
import csv, glob
kf = sorted(glob.glob("$O/tl/**/*kernel_trace.csv", recursive=True))[0]
mf = sorted(glob.glob("$O/tl/**/*memory_copy_trace.csv", recursive=True))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s %s" % (r["Queue_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:28])) for r in csv.DictReader(open(kf))]
M = []
if mf:
    rows = list(csv.DictReader(open(mf[0])))
    print("copy columns:", list(rows[0].keys()))
    M = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))) for r in rows]
E = sorted(K + M)
packs = [i for i, e in enumerate(E) if "pack_scans" in e[2]]
i0, i1 = packs[-3], packs[-2]
t0 = E[i0][0]
for e in E[i0:i1 + 2]:
    d = (e[1] - e[0]) / 1e3
    if d > 15 or "COPY" in e[2] or "pack" in e[2] or "gseg_bin" in e[2] or "height_fit" in e[2]:
        print("%9.1f +%7.1f us  %s" % ((e[0] - t0) / 1e3, d, e[2]))
P
