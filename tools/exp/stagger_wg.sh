#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for W in 1 2; do
  echo -n "raycast-wg $W: "; timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --raycast-wg $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('step %.4f ms %.1f M pts/s raycast in kernel_ms %.4f' % (d['ms_per_step'], d['value']/1e6, d['kernel_ms']['raycast']))"
done; done
for M in "" "--mode p2l"; do echo -n "20 steps $M: "; timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $M 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('step %.4f ms %.1f M pts/s' % (d['ms_per_step'], d['value']/1e6))"; done
