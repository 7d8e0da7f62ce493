"""Two scans per workgroup (slam_icp_params::pair_scans) against one: event-timed registration of config 2's batch
(30 iterations fixed) for several batch sizes, and agreement of the results.  python tools/pair_time.py [S ...]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
m_ga, m_nga = synth.make_map()
st = api.Stream()
for S in sizes:
    batch = synth.make_batch(S)
    d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
    d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
    d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
    d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
    d_pose = api.DeviceArray(d_pose0.shape, np.float64)
    d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
    d_res = api.DeviceArray((S,), api.RESULT_DTYPE)
    ref = None
    for pair in (-1, 1, 2):
        icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=pair)
        ev = [api.Event() for _ in range(2)]
        ms = []
        for k in range(12):
            d_pose.copy_from(d_pose0, st)
            ev[0].record(st)
            icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, st)
            ev[1].record(st)
            st.synchronize()
            if k >= 2:
                ms.append(ev[0].elapsed_ms(ev[1]))
        res, t, R = d_res.download(), d_t.download(), d_R.download()
        if ref is None:
            ref = (res, t, R)
        same_n = bool(np.array_equal(res["n_corr"], ref[0]["n_corr"])) and bool((res["iters"] == 30).all())
        print("S=%4d pair=%2d: %.4f ms (min %.4f)  per 256 scans %.4f ms   n_corr equal %s  |dt| %.2e |dR| %.2e  err vs truth %.4f m"
              % (S, pair, np.mean(ms), np.min(ms), np.mean(ms) * 256 / S, same_n, np.abs(t - ref[1]).max(), np.abs(R - ref[2]).max(),
                 np.abs(t - batch.true_poses[:, :2]).max()), flush=True)
        icp.close()
