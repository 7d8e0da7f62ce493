#!/bin/bash
# quick parameter sweeps on the GPU box (scratch helper; output under gpurun_out/)
mkdir -p gpurun_out
for L in 1 2 4 8 16 32 64; do
  timeout -k 10 120 python bench.py --no-torch --steps 10 --warmup 2 --no-cpu-baseline --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes',$L, d['kernel_ms'])"
done
timeout -k 10 120 python bench.py --no-torch --steps 10 --warmup 2 --no-cpu-baseline --raycast global 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('raycast global', d['kernel_ms'])"
