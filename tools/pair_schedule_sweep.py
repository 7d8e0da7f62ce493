"""Hand-over schedule of the pair kernel (slam_icp_params::first_iterations, far_div): 1024 scans in pairs, event-timed."""
import itertools, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

S = 1024
m_ga, m_nga = synth.make_map()
batch = synth.make_batch(S)
d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
d_pose = api.DeviceArray(d_pose0.shape, np.float64)
d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
st = api.Stream()
for first, far in itertools.product((6, 8, 10, 12, 14), (8, 32, 128)):
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=2, first_iterations=first, far_div=far)
    ev = [api.Event(), api.Event()]
    ms = []
    for k in range(8):
        d_pose.copy_from(d_pose0, st)
        ev[0].record(st)
        icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, None, None, st)
        ev[1].record(st)
        st.synchronize()
        if k >= 2:
            ms.append(ev[0].elapsed_ms(ev[1]))
    print("first_iterations=%2d far_div=%3d: %.4f ms (min %.4f)" % (first, far, np.mean(ms), np.min(ms)), flush=True)
    icp.close()
