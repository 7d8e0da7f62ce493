"""Round 5: the halo lists' pitch against the model's density -- one registration launch (256 scans x 30 iterations) alone on the chip,
one scan per workgroup (fused) and in pairs, for a 5 k / 10 k model, config 2's batch and a chunk of config 5's stream, and
slam_icp_params::list_min_halo.   python tools/exp/halo_forms.py"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth

S = 256
st = api.Stream()
batches = {"config-2 batch": synth.make_batch(S), "config-5 chunk 10": synth.make_batch(S, n_loop=10240, first=10 * S)}
for M in (5000, 10000):
    m_ga, m_nga = synth.make_map(M)
    for bname, batch in batches.items():
        d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
        d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
        d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
        d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
        d_pose = api.DeviceArray(d_pose0.shape, np.float64)
        d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
        d_res = api.DeviceArray((S,), api.RESULT_DTYPE)
        for halo in (-1.0, 0.1, 0.125, 0.15, 0.2, 0.3):
            row = []
            for pair in (-1, 2):
                icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=pair, list_min_halo=halo)
                info = icp.index_info()
                ev = [api.Event() for _ in range(2)]
                ms = []
                for k in range(10):
                    d_pose.copy_from(d_pose0, st)
                    ev[0].record(st)
                    icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, st)
                    ev[1].record(st)
                    st.synchronize()
                    if k >= 2:
                        ms.append(ev[0].elapsed_ms(ev[1]))
                row.append(np.mean(ms))
                icp.close()
            print("M=%5d %-18s min_halo %6.3f -> pitch %.3f halo %.3f list %6d B index %6d B : fused %.4f ms   pairs %.4f ms"
                  % (M, bname, halo, info["list_pitch"], info["list_halo"], info["list_bytes"], info["lds_bytes"], row[0], row[1]), flush=True)
