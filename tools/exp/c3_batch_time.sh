#!/bin/bash
# CCICP::matchSequence: the scene chains as hipGraph replays or call by call, three runs each (round 5)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O/data
python tools/exp/c3_data.py $O/data 50 10
g++ -std=c++17 -O2 -pthread -I include tests/cpp/ccicp_sequence.cpp -o $O/ccicp_sequence -L slam_amd/lib -l:libslam_mi355x.so -Wl,-rpath,$PWD/slam_amd/lib -Wl,-rpath,/opt/rocm/lib
for G in 1 0; do for i in 1 2 3; do
  if [ $G = 0 ]; then export SEQ_NO_GRAPHS=1; else unset SEQ_NO_GRAPHS; fi
  echo -n "graphs $G: "; timeout -k 5 60 $O/ccicp_sequence $O/data 50 10 4 batch 2>&1 | python -c "
import sys,json
t=sys.stdin.read().splitlines()
print(t[0].split('batch: ')[1], '|', json.loads(t[1])['ms_per_match'])"
done; done
