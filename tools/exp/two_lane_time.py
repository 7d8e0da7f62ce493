"""Two registration streams, pair launches back to back, nothing else on the chip: ms per launch and per 256 scans, point-to-point
and point-to-line (what the grid update adds to a launch in the pipelined step is the difference to bench.py's avg_launch_ms).
    python tools/exp/two_lane_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth

S = 256
m_ga, m_nga = synth.make_map()
batch = synth.make_batch(S)
d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
d_R0 = api.DeviceArray.from_host(batch.R, np.float64)
d_t0 = api.DeviceArray.from_host(batch.t, np.float64)
for mode in ("p2p", "p2l"):
    kw = dict(mode=api.ICP_P2L, normals_k=10) if mode == "p2l" else {}
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=2, **kw)
    st = [api.Stream(priority=None), api.Stream(priority=-1)]
    out = [(api.DeviceArray(batch.R.shape, np.float64), api.DeviceArray(batch.t.shape, np.float64)) for _ in range(4)]
    for lanes in (1, 2):
        n = 40
        ev = [(api.Event(), api.Event()) for _ in range(n)]
        for rep in range(2):
            api.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                a = st[k % lanes]
                ev[k][0].record(a)
                icp.fit_batch_from_dev(d_pts, d_off, d_nga, S, d_R0, d_t0, out[k % 4][0], out[k % 4][1], 5.0, None, None, a)
                ev[k][1].record(a)
            api.synchronize()
            dt = time.perf_counter() - t0
        ms = np.mean([a.elapsed_ms(b) for a, b in ev[4:]])
        print("%s, %d stream(s): %.4f ms per launch, %.4f ms per 256 scans over the run" % (mode, lanes, ms, dt / n * 1e3), flush=True)
    icp.close()

# ---- what of the grid update stretches a launch: two registration streams as above, and beside them one grid stream that runs, per
# registration launch, (a) nothing, (b) the raycast of the batch only, (c) finalize_reset only (after one raycast has marked the rows)
grid = api.Grid(2000, 2000, 0.05, rolling=0, min_cluster_points=20, raycast_wg_per_cu=int(os.environ.get("RAYCAST_WG", "1")))
d_R = api.DeviceArray.from_host(batch.R + 0.0, np.float64)
d_t = api.DeviceArray.from_host(batch.true_poses[:, :2].copy(), np.float64)
Rt = np.array([synth.pose_to_Rt(*p)[0].reshape(4) for p in batch.true_poses]); d_R = api.DeviceArray.from_host(Rt, np.float64)
gs = api.Stream(priority=1)
for mode in ("p2p", "p2l"):
    kw = dict(mode=api.ICP_P2L, normals_k=10) if mode == "p2l" else {}
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=2, **kw)
    st = [api.Stream(priority=None), api.Stream(priority=-1)]
    out = [(api.DeviceArray(batch.R.shape, np.float64), api.DeviceArray(batch.t.shape, np.float64)) for _ in range(4)]
    for beside in ("nothing", "raycast", "finalize_reset", "both"):
        n = 40
        ev = [(api.Event(), api.Event()) for _ in range(n)]
        for rep in range(2):
            api.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                a = st[k % 2]
                ev[k][0].record(a)
                icp.fit_batch_from_dev(d_pts, d_off, d_nga, S, d_R0, d_t0, out[k % 4][0], out[k % 4][1], 5.0, None, None, a)
                ev[k][1].record(a)
                if beside in ("raycast", "both"):
                    grid.raycast_scans_dev(d_pts, d_off, S, batch.n_points, d_R, d_t, gs)
                if beside in ("finalize_reset", "both"):
                    if beside == "finalize_reset" and k == 0:
                        grid.raycast_scans_dev(d_pts, d_off, S, batch.n_points, d_R, d_t, gs)
                    grid.finalize(gs) if beside == "finalize_reset" else grid.finalize_reset(gs)
            api.synchronize()
            dt = time.perf_counter() - t0
        ms = np.mean([a.elapsed_ms(b) for a, b in ev[4:]])
        print("%s, two streams, beside them %-14s: %.4f ms per launch, %.4f ms per 256 scans over the run" % (mode, beside, ms, dt / n * 1e3), flush=True)
    icp.close()
