#!/bin/bash
# CCICP::matchSequence against the runtime's hardware-queue count (round 5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O/data
python tools/exp/c3_data.py $O/data 50 10
g++ -std=c++17 -O2 -pthread -I include tests/cpp/ccicp_sequence.cpp -o $O/ccicp_sequence -L slam_amd/lib -l:libslam_mi355x.so -Wl,-rpath,$PWD/slam_amd/lib -Wl,-rpath,/opt/rocm/lib
for Q in default 2 4 8; do for f in seq ahead batch; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  echo -n "GPU_MAX_HW_QUEUES=$Q $f: "; timeout -k 5 60 $O/ccicp_sequence $O/data 50 10 3 $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['ms_per_match'], d['ms_per_cloud_with_target_updates'])"
done; done
