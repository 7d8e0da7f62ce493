#!/bin/bash
# round 5 re-sweep of the pipelined step's knobs on the final kernels: raycast workgroups per CU x grid lag, both solvers
OUT=gpurun_out/knobs.txt
: > $OUT
for rep in 1 2; do
for mode in p2p p2l; do
for wg in 1 2; do
for lag in 2 3 4; do
  v=$(timeout -k 10 150 python3 bench.py --mode $mode --steps 60 --warmup 8 --no-extras --no-cpu-baseline --raycast-wg $wg --grid-lag $lag 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
  echo "$mode raycast-wg $wg grid-lag $lag rep $rep: $v" >> $OUT
done; done; done; done
sort $OUT
