"""Round 5: would COARSE halo lists (pitch 1.53 m, halo 0.38 m: the next lattice of the 10 k model that fits LDS) pay in the EARLY iterations,
where the ring search costs 24-45 us per iteration and the fine lists (0.55 / 0.138 m) cannot certify yet?  One registration launch
(256 scans x 30 iterations) alone on the chip, fused and pairs: fine / coarse lists x the iteration the ring form may hand over at.
python tools/exp/coarse_lists.py"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth

S = 256
st = api.Stream()
m_ga, m_nga = synth.make_map()
batch = synth.make_batch(S)
d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
d_pose = api.DeviceArray(d_pose0.shape, np.float64)
d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
d_res = api.DeviceArray((S,), api.RESULT_DTYPE)
for halo in (0.0, 0.3):
    for first in (1, 2, 3, 4, 6, 10):
        for far_div in (32, 8, 2):
            row = []
            for pair in (-1, 2):
                icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=pair, list_min_halo=halo, first_iterations=first, far_div=far_div)
                info = icp.index_info()
                ev = [api.Event() for _ in range(2)]
                ms = []
                for k in range(8):
                    d_pose.copy_from(d_pose0, st)
                    ev[0].record(st)
                    icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, st)
                    ev[1].record(st)
                    st.synchronize()
                    if k >= 2:
                        ms.append(ev[0].elapsed_ms(ev[1]))
                row.append(np.mean(ms))
                icp.close()
            print("lists %.3f / %.3f m, ring form at least %2d iterations, hand over at <= n/%-2d far: fused %.4f ms  pairs %.4f ms"
                  % (info["list_pitch"], info["list_halo"], first, far_div, row[0], row[1]), flush=True)
