"""One scan at a time against a model too large for LDS (2 x 19 999 points, the CCICP cap): slam_icp_fit through
the per-iteration step + solve launches.  Prints the time per fit."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map(39998)
batch = synth.make_batch(8, n_loop=256)
icp = api.Icp(m_ga, m_nga)                      # reference defaults: 20 iterations, 1e-6
print("index", icp.index_info()["in_lds"])
for rep in range(3):
    t0 = time.perf_counter(); its = []
    for s in range(8):
        ga, nga = batch.scan(s)
        R, t, res = icp.fit(ga, nga, batch.R[s].reshape(2, 2), batch.t[s])
        its.append(res.iters)
    dt = (time.perf_counter() - t0) / 8
    print("single scan vs 40k-point model: %.3f ms per fit, %.1f iterations" % (dt * 1e3, np.mean(its)))
