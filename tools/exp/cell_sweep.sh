#!/bin/bash
OUT=gpurun_out/cell_sweep.txt
: > $OUT
for rep in 1 2 3; do
  for cell in 0 0.5 0.6 0.75 0.9; do
    v=$(timeout -k 10 120 python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --cell $cell 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "cell $cell rep $rep ms_per_step $v" >> $OUT
  done
done
sort $OUT
