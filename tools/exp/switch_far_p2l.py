"""P2P: the hand-over guard (far_div: hand over once at most n / far_div queries are beyond the lists' certified radius) against the first
iteration it may happen at; config 2's scans, 512 in pairs, one launch alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map()
st = api.Stream()
S, pair = 512, 2
P2L = dict(mode=api.ICP_P2L, normals_k=10)
batch = synth.make_batch(S)
d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
d_pose = api.DeviceArray(d_pose0.shape, np.float64)
d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
d_res = api.DeviceArray((S,), api.RESULT_DTYPE)
ref = None
for rep in range(2):
    for far in (128, 32, 16, 8):
        for k in (6, 4, 3, 2, 1):
            icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=pair, first_iterations=k, far_div=far, **P2L)
            ev = [api.Event() for _ in range(2)]
            ms = []
            for r in range(10):
                d_pose.copy_from(d_pose0, st)
                ev[0].record(st)
                icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, st)
                ev[1].record(st)
                st.synchronize()
                if r >= 2:
                    ms.append(ev[0].elapsed_ms(ev[1]))
            res, t = d_res.download(), d_t.download()
            if ref is None:
                ref = (res, t)
            if rep:
                print("far_div=%2d first_iterations=%2d: %.4f ms (min %.4f)  n_corr equal %s  |dt| %.1e"
                      % (far, k, np.mean(ms), np.min(ms), bool(np.array_equal(res["n_corr"], ref[0]["n_corr"])), np.abs(t - ref[1]).max()), flush=True)
            icp.close()
