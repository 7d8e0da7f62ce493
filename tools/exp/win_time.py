"""One registration launch alone (512 scans in pairs, 256 one per workgroup), both solvers, library defaults: for A/B of library builds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map()
st = api.Stream()
out = []
for mode in ("p2p", "p2l"):
    for S, pair in ((512, 2), (256, -1)):
        batch = synth.make_batch(S)
        d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
        d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
        d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
        d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
        d_pose = api.DeviceArray(d_pose0.shape, np.float64)
        d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
        d_res = api.DeviceArray((S,), api.RESULT_DTYPE)
        kw = dict(max_iter=30, min_delta=-1.0, pair_scans=pair)
        if mode == "p2l":
            kw.update(mode=api.ICP_P2L, normals_k=10)
        icp = api.Icp(m_ga, m_nga, **kw)
        ev = [api.Event() for _ in range(2)]
        ms = []
        for r in range(14):
            d_pose.copy_from(d_pose0, st)
            ev[0].record(st)
            icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, st)
            ev[1].record(st)
            st.synchronize()
            if r >= 4:
                ms.append(ev[0].elapsed_ms(ev[1]))
        out.append("%s/%d/%s %.4f" % (mode, S, "pairs" if pair == 2 else "one", np.mean(ms)))
        icp.close()
print("  ".join(out))
