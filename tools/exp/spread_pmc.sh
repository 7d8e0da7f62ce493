#!/bin/bash
# instruction / wait counters of the spread kernel (tools/spread_time.py's launches), one rocprofv3 pass per counter group
R=$PWD; mkdir -p $R/gpurun_out/r06; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/r06/pmc_$i -o p -- python3 $R/tools/spread_time.py 3 > /dev/null 2>&1 || echo "pass $i failed"
done
ls $R/gpurun_out/r06/pmc_*/
