#!/bin/bash
# config 3's batch form with its scene chains replayed as hipGraphs (default until round 6) against call by call (SEQ_NO_GRAPHS=1)
cd $GRAFT_REPO_ROOT
rm -rf /tmp/c3b && SLAM_C3_KEEP_DIR=/tmp/c3b python3 tools/bench_config3.py 50 > /dev/null 2>&1
for rep in 1 2 3; do
  for g in graphs calls; do
    if [ $g = calls ]; then export SEQ_NO_GRAPHS=1; else unset SEQ_NO_GRAPHS; fi
    /tmp/c3b/ccicp_sequence /tmp/c3b 50 10 4 batch 2>&1 | python3 -c "
import sys,json
L=sys.stdin.read().splitlines()
d=json.loads([l for l in L if l.startswith('{')][-1])
print('$g rep $rep', d['ms_per_match'], [l for l in L if l.startswith('matchSequence')][-1][:140])"
  done
done
