#!/bin/bash
# usage: bash tools/pmc_model.sh TAG MAP_POINTS [KIND]  -- counter passes of the registration launch against a model of MAP_POINTS
# points (one stream, one scan per workgroup: the kernel alone on the chip), into gpurun_out/pmcm_TAG_*; summary in
# gpurun_out/pmcm_TAG.txt (tools/pmc_summary.py).  Counter passes use --pmc with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; M=${2:-39998}; KIND=${3:-room}
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "FETCH_SIZE"; do
  i=$((i+1)); rm -rf gpurun_out/pmcm_${TAG}_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d gpurun_out/pmcm_${TAG}_$i -- python3 bench.py --no-torch --map-points $M --map-kind $KIND --no-pipeline --no-graph --steps 3 --warmup 1 --no-extras --no-cpu-baseline $EXTRA > gpurun_out/pmcm_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py "gpurun_out/pmcm_${TAG}_*/*/*counter_collection.csv" > gpurun_out/pmcm_${TAG}.txt 2>&1
grep -A 40 "icp_fit" gpurun_out/pmcm_${TAG}.txt | head -60
