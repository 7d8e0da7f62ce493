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
grid = api.Grid(2000, 2000, 0.05, rolling=0, min_cluster_points=20, raycast_seg_items=int(os.environ.get('X_SEG', 0)),
                raycast_wg_per_cu=int(os.environ.get('X_WG', 0)))
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


def run_pairs(n_icp_streams, pair):
    """registration with two scans per workgroup (half the CUs per batch) on n alternating streams, the grid on another"""
    icp2 = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=pair)
    A = [api.Stream(-1 if i == 0 else None) for i in range(n_icp_streams)]
    b = api.Stream(1)
    NB = 2 * n_icp_streams
    poses = [api.DeviceArray(d_pose0.shape, np.float64) for _ in range(NB)]
    R_ = [p.view(0, batch.R.shape) for p in poses]
    t_ = [p.view(batch.R.size, batch.t.shape) for p in poses]
    icp_done = [api.Event() for _ in range(NB)]
    grid_done = [api.Event() for _ in range(NB)]
    def steps(n):
        for k in range(n):
            s, a = k % NB, A[k % n_icp_streams]
            a.wait_event(grid_done[s])
            poses[s].copy_from(d_pose0, a)
            icp2.fit_batch_dev(d_pts, d_off, d_nga, S, R_[s], t_[s], 5.0, None, None, a)
            icp_done[s].record(a)
            b.wait_event(icp_done[s])
            grid.reset_counts(b)
            grid.raycast_scans_dev(d_pts, d_off, S, P, R_[s], t_[s], b)
            grid.finalize(b)
            grid_done[s].record(b)
    for e in grid_done:
        e.record(b)
    steps(8)
    api.synchronize()
    t0 = time.perf_counter()
    steps(K)
    api.synchronize()
    dt_ = (time.perf_counter() - t0) / K * 1e3
    icp2.close()
    return dt_


print("one stream, call by call: %.4f ms/step" % run_one())
for n, pair in ((2, 2), (2, -1), (3, 2), (1, 2)):
    print("%d registration streams, pair_scans=%d: %.4f ms/step" % (n, pair, run_pairs(n, pair)))
for pa, pb, name in ((None, None, "two streams, default priorities"), (1, -1, "ICP high, grid low"), (-1, 1, "ICP low, grid high")):
    print("%s: %.4f ms/step" % (name, run(pa, pb)))
h, m = grid.read_counts()
print("counts", int(h.sum()), int(m.sum()))
