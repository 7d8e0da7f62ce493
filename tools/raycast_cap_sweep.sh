#!/bin/bash
# the pipelined step against the number of persistent raycast workgroups (slam_grid_params::raycast_max_workgroups)
for cap in 0 384 256 192 128 96 64; do for steps in 20 50; do
  python bench.py --no-extras --no-cpu-baseline --steps $steps --raycast-max-wg $cap 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap=$cap steps=$steps', round(d['ms_per_step'],4), round(d['value']/1e6,1), d['kernel_ms'])"
done; done
