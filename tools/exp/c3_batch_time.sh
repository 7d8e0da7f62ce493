#!/bin/bash
# CCICP::matchSequence: the host's clock of the batch phases, eight runs (round 5: the form is bimodal run to run)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O/data
python tools/exp/c3_data.py $O/data 50 10
g++ -std=c++17 -O2 -pthread -I include tests/cpp/ccicp_sequence.cpp -o $O/ccicp_sequence -L slam_amd/lib -l:libslam_mi355x.so -Wl,-rpath,$PWD/slam_amd/lib -Wl,-rpath,/opt/rocm/lib
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 5 60 $O/ccicp_sequence $O/data 50 10 4 batch 2>&1 | python -c "
import sys,json
t=sys.stdin.read().splitlines()
print(t[0].split('batch: ')[1], '|', json.loads(t[1])['ms_per_match'])"
done
