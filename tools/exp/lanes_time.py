"""Registration launches of 256 scans back to back on 1..4 streams, nothing else on the chip: ms per 256 scans.
    [SLAM_ICP_TEAMS=4] python tools/exp/lanes_time.py"""
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
icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=2)
st = [api.Stream(private_queue=True) for _ in range(6)]
out = [(api.DeviceArray(batch.R.shape, np.float64), api.DeviceArray(batch.t.shape, np.float64)) for _ in range(8)]
for lanes in (1, 2, 3, 4, 6):
    n = 48
    ev = [(api.Event(), api.Event()) for _ in range(n)]
    for rep in range(2):
        api.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            a = st[k % lanes]
            ev[k][0].record(a)
            icp.fit_batch_from_dev(d_pts, d_off, d_nga, S, d_R0, d_t0, out[k % 8][0], out[k % 8][1], 5.0, None, None, a)
            ev[k][1].record(a)
        api.synchronize()
        dt = time.perf_counter() - t0
    ms = np.mean([a.elapsed_ms(b) for a, b in ev[8:]])
    print("teams %s, %d stream(s): %.4f ms per launch, %.4f ms per 256 scans over the run" % (os.environ.get("SLAM_ICP_TEAMS", "2"), lanes, ms, dt / n * 1e3), flush=True)
