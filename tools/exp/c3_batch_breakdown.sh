#!/bin/bash
# config 3's batch form (CCICP::matchSequence): the host's clock per batch -- scene chains enqueued | everything enqueued | results back
cd $GRAFT_REPO_ROOT
rm -rf /tmp/c3b && SLAM_C3_KEEP_DIR=/tmp/c3b python3 tools/bench_config3.py 20 > /dev/null 2>&1
/tmp/c3b/ccicp_sequence /tmp/c3b 20 10 6 batch 2>&1 | tail -3
SEQ_NO_GRAPHS=1 /tmp/c3b/ccicp_sequence /tmp/c3b 20 10 6 batch 2>&1 | tail -2
