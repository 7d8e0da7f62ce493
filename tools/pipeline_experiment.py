#!/usr/bin/env python3
"""Config 2's step as a two-stream pipeline over consecutive batches: the registration of batch k+1 beside the grid update
of batch k (independent: the ICP does not touch the planes), with the streams' priorities either way round.  Prints
ms per step in steady state against the one-stream step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

S, K = 256, 60
m_ga, m_nga = synth.make_map()
batch = synth.make_batch(S)
P = batch.n_points
icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0)
grid = api.Grid(2000, 2000, 0.05, rolling=0, min_cluster_points=20)
d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
pose = [api.DeviceArray(d_pose0.shape, np.float64) for _ in range(2)]
dR = [p.view(0, batch.R.shape) for p in pose]
dt = [p.view(batch.R.size, batch.t.shape) for p in pose]


def run(pa, pb):
    a, b = api.Stream(pa), api.Stream(pb)
    icp_done = [api.Event() for _ in range(2)]
    grid_done = [api.Event() for _ in range(2)]
    def steps(n):
        for k in range(n):
            s = k % 2
            a.wait_event(grid_done[s])
            pose[s].copy_from(d_pose0, a)
            icp.fit_batch_dev(d_pts, d_off, d_nga, S, dR[s], dt[s], 5.0, None, None, a)
            icp_done[s].record(a)
            b.wait_event(icp_done[s])
            grid.reset_counts(b)
            grid.raycast_scans_dev(d_pts, d_off, S, P, dR[s], dt[s], b)
            grid.finalize(b)
            grid_done[s].record(b)
    for e in grid_done:
        e.record(b)
    steps(6)
    api.synchronize()
    t0 = time.perf_counter()
    steps(K)
    api.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def run_one():
    a = api.Stream()
    def steps(n):
        for k in range(n):
            pose[0].copy_from(d_pose0, a)
            icp.fit_batch_dev(d_pts, d_off, d_nga, S, dR[0], dt[0], 5.0, None, None, a)
            grid.reset_counts(a)
            grid.raycast_scans_dev(d_pts, d_off, S, P, dR[0], dt[0], a)
            grid.finalize(a)
    steps(6)
    api.synchronize()
    t0 = time.perf_counter()
    steps(K)
    api.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


print("one stream, call by call: %.4f ms/step" % run_one())
for pa, pb, name in ((None, None, "two streams, default priorities"), (1, -1, "ICP high, grid low"), (-1, 1, "ICP low, grid high")):
    print("%s: %.4f ms/step" % (name, run(pa, pb)))
h, m = grid.read_counts()
print("counts", int(h.sum()), int(m.sum()))
