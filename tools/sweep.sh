#!/bin/bash
# quick parameter sweeps on the GPU box (scratch helper; output under gpurun_out/)
run() { timeout -k 10 120 python bench.py --no-torch --steps 10 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', {k: round(v,4) for k,v in d['kernel_ms'].items()}, round(d['value']/1e6,1))"; }
for L in 1 2 4; do run "lanes=$L" "--lanes $L"; done
