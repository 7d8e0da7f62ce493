#!/usr/bin/env python3
"""bench.py's endpoint leg alone (MLS::addToOccupancy's endpoint update, mls.cpp:73-142, through slam_grid_add_endpoints_dev +
slam_grid_finalize_reset on config 2's registered endpoints and on a segmented config-3 cloud), for the profiler:
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_endpoints.py
prints the leg's JSON."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from slam_amd import api, synth

api.set_device(0)
m_ga, m_nga = synth.make_map(bench.MAP_POINTS)
batch = synth.make_batch(bench.N_SCANS)
icp = api.Icp(m_ga, m_nga, max_iter=bench.N_ITERS, min_delta=-1.0)
R, t, res, _ = icp.fit_batch(batch, indist=5.0)
icp.close()
# (without the leg's 8 M-point input: the counters below are means per launch of the two BASELINE inputs)
print(json.dumps(bench.endpoint_leg(api, synth, batch, R, t, bench.GRID, bench.RES, False, at_size=False)))
