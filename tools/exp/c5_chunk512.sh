#!/bin/bash
# config 5 with 512-scan chunks: one scan per workgroup (the sliding target's default) against pairs (round 5)
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; timeout -k 10 200 python bench.py --config 5 --stream-scans 10240 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); print('%.4f ms per chunk  %.1f M pts/s  %s' % (d['ms_per_step'], d['value']/1e6, d['config']['mapper']))"; }
for rep in 1 2; do
run --chunk 512 --window 2 --rebuild-every 2 --merge-every 4
run --chunk 512 --window 2 --rebuild-every 2 --merge-every 4 --pair-scans 2
run --chunk 512 --window 4 --rebuild-every 4 --merge-every 4 --pair-scans 2
run --chunk 256
done
