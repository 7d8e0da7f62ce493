#!/bin/bash
# usage: tools_pmc_run.sh TAG  (env vars pass through); collects three PMC passes into gpurun_out/pmc_TAG_*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ITEMS SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_${TAG}_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python bench.py --no-torch --steps 3 --warmup 1 --no-cpu-baseline --lanes ${LANES:-0} > gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
