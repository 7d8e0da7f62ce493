#!/bin/bash
OUT=gpurun_out/live_sweep.txt
: > $OUT
for rep in 1 2 3 4 5; do
  for live in 1 0; do
    v=$(SLAM_BENCH_LIVE=$live timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --start-stagger-us 150 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "live $live rep $rep ms_per_step $v" >> $OUT
    v=$(SLAM_BENCH_LIVE=$live timeout -k 10 120 python3 bench.py --gpus 1 --steps 100 --warmup 5 --no-extras --no-cpu-baseline --start-stagger-us 150 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "live $((live+10)) rep $rep ms_per_step $v" >> $OUT
  done
done
python3 - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open("gpurun_out/live_sweep.txt"):
    w=l.split(); d[int(w[1])].append(float(w[5]))
for k in sorted(d): print(k, " ".join("%.4f"%x for x in sorted(d[k])), " median %.4f" % sorted(d[k])[len(d[k])//2])
PY
