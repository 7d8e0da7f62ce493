#!/bin/bash
# the grid update's streams confined to 32 - K CUs per XCD (bench.py --grid-cu-cap K): pipelined step, 60 steps
OUT=gpurun_out/grid_cus.txt
: > $OUT
for rep in 1 2; do
for K in 0 8 16 20 24 28; do
  v=$(timeout -k 10 150 python3 bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline --grid-cu-cap $K 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  launch %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "grid-cu-cap $K: $v" >> $OUT
done
done
cat $OUT
