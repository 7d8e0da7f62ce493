#!/bin/bash
# the pipelined step's launch settings again, with the staggered raycast (round 5)
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('step %.4f ms %.1f M pts/s launch %.4f' % (d['ms_per_step'], d['value']/1e6, d['roofline']['avg_launch_ms']))"; }
for rep in 1 2; do
run
run --step-streams 3
run --step-streams 4
run --grid-lag 2
run --grid-lag 4
run --start-stagger-us 0
run --one-grid
done
