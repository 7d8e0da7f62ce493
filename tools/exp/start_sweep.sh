#!/bin/bash
# Runs ON THE GPU BOX: the driver's own bench command (20 steps, 5 warm-up) against the host-side stagger between the first two
# registration launches of a run; five runs of each setting, ms per step.
OUT=gpurun_out/start_sweep.txt
: > $OUT
for rep in 1 2 3 4 5; do
  for us in 0 100 150 200 250 300 400; do
    v=$(timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --start-stagger-us $us 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "stagger_us $us rep $rep ms_per_step $v" >> $OUT
  done
done
python3 - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open("gpurun_out/start_sweep.txt"):
    w=l.split(); d[int(w[1])].append(float(w[5]))
for k in sorted(d): print(k, " ".join("%.4f"%x for x in sorted(d[k])), " median %.4f" % sorted(d[k])[len(d[k])//2])
PY
