#!/bin/bash
# pitch of the ring search's cell index in the default (ring search, then list sweeps) schedule
for C in 0 0.33 0.36 0.4 0.45 0.5 0.6; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --cell $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cell=$C', round(d['config']['icp_index']['cell'],3), d['config']['icp_index']['lds_bytes'], {k: round(v,4) for k,v in d['kernel_ms'].items()}, round(d['value']/1e6,1))" || echo "cell=$C failed"
done
