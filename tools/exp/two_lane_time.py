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
