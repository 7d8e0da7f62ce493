#!/bin/bash
# CCICP::matchSequence: lanes (streams the scene chains are dealt over) x eight runs each (round 5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O/data
python tools/exp/c3_data.py $O/data 50 10
for L in 4 8 2; do
g++ -std=c++17 -O2 -pthread -DSLAM_CCICP_SEQ_LANES=$L -I include tests/cpp/ccicp_sequence.cpp -o $O/ccicp_sequence_$L -L slam_amd/lib -l:libslam_mi355x.so -Wl,-rpath,$PWD/slam_amd/lib -Wl,-rpath,/opt/rocm/lib
echo -n "lanes $L:"
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 5 60 $O/ccicp_sequence_$L $O/data 50 10 4 batch 2>/dev/null | python -c "
import sys,json
print(' %.3f' % json.loads(sys.stdin.read().splitlines()[-1])['ms_per_match'], end='')"
done; echo; done
