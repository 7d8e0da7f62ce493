#!/bin/bash
# Kernel trace of the default bench on the GPU box: duration of the two ICP launches of each step and the gap
# between them (the hand-over is a barrier over the whole batch).  usage: bash tools/prof2p.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof2p && mkdir -p gpurun_out/prof2p
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-torch > gpurun_out/prof2p/out.json 2> gpurun_out/prof2p/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof2p/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'icp_fit' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for a, b in list(zip(rows[::2], rows[1::2]))[-4:]:
    print('A %.1f us  gap %.1f us  B %.1f us   LDS %s / %s' % ((int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3, (int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3, (int(b['End_Timestamp'])-int(b['Start_Timestamp']))/1e3, a.get('LDS_Block_Size',''), b.get('LDS_Block_Size','')))
PY
