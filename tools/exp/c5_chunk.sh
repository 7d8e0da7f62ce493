#!/bin/bash
# config 5 against the chunk size (256 scans per chunk is bench.py's; BASELINE config 5 names none): from two scans per CU on a chunk registers in pairs (round 5)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for C in 256 512 1024; do for RE in 4 2; do
  echo -n "chunk $C rebuild-every $RE: "; timeout -k 10 200 python bench.py --config 5 --stream-scans 10240 --chunk $C --rebuild-every $RE --merge-every $((2048 / C)) 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); print('%.4f ms per chunk = %.4f per 256 scans  %.1f M pts/s  err %.4f  %s' % (d['ms_per_step'], d['ms_per_step']*256/$C, d['value']/1e6, d['max_pose_error_m'], d['config']['mapper']))"
done; done; done
