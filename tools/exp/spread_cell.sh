#!/bin/bash
# config 3's fit (spread form, tile form) against the pitch of the target's cell index
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/spread_cell.txt; : > $OUT
for C in "" 0.075 0.1 0.2 0.3 0.45; do
  SLAM_SPREAD_CELL=$C timeout -k 10 120 python tools/spread_time.py 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1])
print('cell=$C', d['config3_index']['cell'], d['config3_index']['nx'], d['config3_index']['ny'], [d['config3_cloud%d'%k]['us_per_fit_median'] for k in (1,5,9)], [d['config3_cloud%d'%k]['iterations'] for k in (1,5,9)])" >> $OUT || echo "cell=$C failed" >> $OUT
done
cat $OUT
