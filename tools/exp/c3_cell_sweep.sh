#!/bin/bash
# config 3: the pitch of the target's cell index (20 870 points, hundreds per cell near the sensor) against the one-scan fit (round 5)
cd $GRAFT_REPO_ROOT
for C in 0 0.05 0.075 0.1 0.2 0.3; do
  timeout -k 10 150 python tools/bench_config3.py 20 $C 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1])
print('cell=$C', 'chain ms/cloud', d['ms_per_cloud_chain'], 'stepwise', d['ms_per_cloud'], 'iters', round(d['mean_icp_iterations'],2), 'index build', d['target_index_build_ms'], 'err', round(d['mean_xy_error_m'],4))" || echo "cell=$C failed"
done
