#!/bin/bash
# lanes-per-point sweep of icp_fit_kernel on config 2; runs on the GPU box, output on stdout
run() { timeout -k 10 120 python bench.py --no-torch --steps 10 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', {k: round(v,4) for k,v in d['kernel_ms'].items()}, round(d['value']/1e6,1))" || exit 1; }
for L in 0 -2 -1 1 2 4 8 16 64; do run "lanes=$L" "--lanes $L"; done
