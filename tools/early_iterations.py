"""Cost of the first iterations of the ring-search launch as a function of the cell pitch and the lanes per
point: one fit_batch of 256 scans with max_iter 2, 4, 10, timed with HIP events."""
import sys; sys.path.insert(0, ".")
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map(); batch = synth.make_batch(256)
d_pts = api.DeviceArray.from_host(batch.pts, np.float64); d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
d_R0 = api.DeviceArray.from_host(batch.R, np.float64); d_t0 = api.DeviceArray.from_host(batch.t, np.float64)
d_R = api.DeviceArray(batch.R.shape, np.float64); d_t = api.DeviceArray(batch.t.shape, np.float64)
d_res = api.DeviceArray((256,), api.RESULT_DTYPE)
for iters in (2, 4, 10):
    for cell in (0.0, 0.45, 0.6, 0.9, 1.2):
        for lanes in (2, 4):
            icp = api.Icp(m_ga, m_nga, max_iter=iters, min_delta=-1.0, lanes_per_point=lanes, cell_size=cell)
            e0, e1 = api.Event(), api.Event()
            ms = []
            for rep in range(6):
                d_R.copy_from(d_R0); d_t.copy_from(d_t0)
                e0.record(); icp.fit_batch_dev(d_pts, d_off, d_nga, 256, d_R, d_t, 5.0, d_res); e1.record(); e1.synchronize()
                ms.append(e0.elapsed_ms(e1))
            print("iters %2d cell %.2f lanes %d: %.1f us (%.1f us/iteration)" % (iters, icp.index_info()["cell"], lanes, np.median(ms[2:]) * 1e3, np.median(ms[2:]) * 1e3 / iters))
            icp.close()
