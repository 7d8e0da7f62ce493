#!/usr/bin/env python3
"""slam_icp_create in point-to-line mode (the normals by k-NN on the cell index, icp_build.hip) against model size and cell pitch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth

for npts in (10000, 20000, 30000, 39998):
    ga, nga = synth.make_map(npts)
    for cell in (0.0, 0.06, 0.03):
        kw = dict(mode=api.ICP_P2L, normals_k=10)
        if cell:
            kw["cell_size"] = cell
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            icp = api.Icp(ga, nga, **kw)
            ts.append(time.perf_counter() - t0)
            info = icp.index_info()
            icp.close()
        t0 = time.perf_counter()
        icp = api.Icp(ga, nga)
        tp = time.perf_counter() - t0
        icp.close()
        print("%d points, cell %s -> %.3f m (%d x %d, in_lds %s): P2L create %.3f ms (min %.3f); P2P create %.3f" %
              (npts, cell or "default", info["cell"], info["nx"], info["ny"], info["in_lds"], np.median(ts) * 1e3, min(ts) * 1e3, tp * 1e3))

# the bench's cap model: 19 999 Gaussian + 19 999 uniform points over 80 x 60 m (scattered, with stragglers in the fringe)
rs = np.random.RandomState(7)
big = (rs.randn(19999, 2) * [30.0, 20.0], rs.rand(19999, 2) * [80.0, 60.0] - [40.0, 30.0])
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    icp = api.Icp(big[0], big[1], mode=api.ICP_P2L, normals_k=10)
    ts.append(time.perf_counter() - t0)
    icp.close()
print("scattered 2 x 19999: P2L create %.3f ms (min %.3f)" % (np.median(ts) * 1e3, min(ts) * 1e3))
