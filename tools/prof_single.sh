#!/bin/bash
# Kernel trace of the single-scan path (config-3 tool, 6 clouds): step / solve kernel durations and gaps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_single && mkdir -p gpurun_out/prof_single
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -- python3 tools/bench_config3.py 6 > gpurun_out/prof_single/out.json 2> gpurun_out/prof_single/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_single/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
f = glob.glob('gpurun_out/prof_single/*/*kernel_trace.csv')[0]
rows = sorted([r for r in csv.DictReader(open(f)) if 'icp_step' in r['Kernel_Name'] or 'icp_solve' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))
for a, b in list(zip(rows, rows[1:]))[200:212]:
    print(a['Kernel_Name'][:40], (int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3, 'us, gap to next', (int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3)
PY
