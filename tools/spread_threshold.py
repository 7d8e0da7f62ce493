#!/usr/bin/env python3
"""ICP time of a batch of S scans (1081 beams, 10 k-point map, 30 iterations) in the two forms: one workgroup per scan
(spread_scans = -1) and every scan spread over n_cu / S workgroups (spread_scans = S).  Picks kSpreadMinParts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

m_ga, m_nga = synth.make_map()
for S in (1, 2, 4, 8, 16, 32, 64, 128):
    batch = synth.make_batch(S, n_loop=256)
    row = []
    for spread in (-1, S):
        icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, spread_scans=spread)
        d = [api.DeviceArray.from_host(a, dt) for a, dt in ((batch.pts, np.float64), (batch.scan_off, np.int32), (batch.scan_nga, np.int32))]
        d_R0, d_t0 = api.DeviceArray.from_host(batch.R, np.float64), api.DeviceArray.from_host(batch.t, np.float64)
        d_R, d_t = api.DeviceArray(batch.R.shape, np.float64), api.DeviceArray(batch.t.shape, np.float64)
        st = api.Stream()
        e0, e1 = api.Event(), api.Event()
        ts = []
        for k in range(12):
            d_R.copy_from(d_R0, st); d_t.copy_from(d_t0, st)
            e0.record(st)
            icp.fit_batch_dev(d[0], d[1], d[2], S, d_R, d_t, 5.0, None, None, st)
            e1.record(st)
            st.synchronize()
            ts.append(e0.elapsed_ms(e1))
        row.append(np.median(ts[2:]))
        icp.close()
    print("S=%3d  one workgroup per scan %.3f ms   spread %.3f ms   (%.1f / %.1f M points/s)"
          % (S, row[0], row[1], batch.n_points / row[0] / 1e3, batch.n_points / row[1] / 1e3))
