"""Diagnostic (measure build, SLAM_ICP_STAMPS=1): queries per iteration whose certificate failed, list form, late iterations."""
import os, sys, ctypes as C
os.environ["SLAM_ICP_STAMPS"] = "1"; os.environ["SLAM_AMD_MEASURE"] = "1"
sys.path.insert(0, ".")
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map(); batch = synth.make_batch(256)
for iters in (12, 16, 20, 30):
    icp = api.Icp(m_ga, m_nga, max_iter=iters, min_delta=-1.0)
    icp.fit_batch(batch)
    out = (C.c_double * 9)()
    L = api.lib(); L.slam_icp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    api.check(L.slam_icp_debug_stamps(icp.h, out))
    v = np.array(out[:]); late = max(iters - 6, 1)
    print("iters %d: per iteration per wave search %.0f reduce %.0f barrier %.0f solve %.0f | late search %.0f coop %.0f | fell_back sum per wave per late iteration %.2f (x16 waves /1000 = failed per iteration)"
          % ((iters,) + tuple(v[:4] / iters) + (v[6] / late, v[7] / late, v[4] / late)))
