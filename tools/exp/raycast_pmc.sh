#!/bin/bash
# counters of raycast_tiled_kernel alone (tools/raycast_time.py): LDS conflicts, VALU busy and lanes (round 5: after the staggered walk)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/rpmc; mkdir -p gpurun_out/rpmc
i=0
for PMC in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d gpurun_out/rpmc/p$i -- python3 tools/raycast_time.py wg_per_cu=${1:-1} > gpurun_out/rpmc/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/rpmc/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "raycast_tiled" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k,v in acc.items()}
for k in sorted(m): print("%-24s %.4g" % (k, m[k]))
cyc=m["GRBM_GUI_ACTIVE"]/8
print("kernel cycles %.0f; VALU busy %.1f %% of all SIMD cycles; active lanes %.1f %%; LDS bank-conflict share %.1f %%; LDS busy (SQ_ACTIVE_INST_LDS*4/(1024*cyc)) %.1f %%; wait-for-LDS share of wave cycles %.1f %%"
      % (cyc, 400*m["SQ_ACTIVE_INST_VALU"]/(1024*cyc), 100*m["SQ_THREAD_CYCLES_VALU"]/(64*m["SQ_ACTIVE_INST_VALU"]), 100*m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"],
         400*m["SQ_ACTIVE_INST_LDS"]/(1024*cyc), 100*m["SQ_WAIT_INST_LDS"]/m["SQ_WAVE_CYCLES"]))
PY
