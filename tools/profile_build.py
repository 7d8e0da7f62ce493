#!/usr/bin/env python3
"""slam_icp_create / slam_icp_fit in a loop for rocprofv3 --kernel-trace --stats: the device build of the model index
(10 k-point map, 2 x 19 999-point map) and the single-scan fit against either."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

rs = np.random.RandomState(3)
m_ga, m_nga = synth.make_map()
big = synth.make_map(39998)
batch = synth.make_batch(1, n_loop=256)
t_ga, t_nga = batch.scan(0)
for name, (ga, nga) in (("10k", (m_ga, m_nga)), ("2x19999", big)):
    api.Icp(ga, nga).close()
    tc, tf = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        icp = api.Icp(ga, nga, max_iter=20, min_delta=-1.0)
        t1 = time.perf_counter()
        icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
        t2 = time.perf_counter()
        tc.append(t1 - t0); tf.append(t2 - t1)
        info = icp.build_info()[1]
        icp.close()
    # ... and the handle kept, as a caller with a fixed map keeps it (the first fit of a handle reserves its buffers)
    icp = api.Icp(ga, nga, max_iter=20, min_delta=-1.0)
    tk = []
    for k in range(220):
        t1 = time.perf_counter()
        icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
        if k >= 20:
            tk.append(time.perf_counter() - t1)
    icp.close()
    print("%s: create %.3f ms (min %.3f)  parts %s   fit(1081 pts, 20 it, host API) first of a handle %.3f ms (min %.3f), handle kept %.3f ms (min %.3f)"
          % (name, np.median(tc) * 1e3, min(tc) * 1e3, [round(x, 3) for x in info], np.median(tf) * 1e3, min(tf) * 1e3,
             np.median(tk) * 1e3, min(tk) * 1e3))
