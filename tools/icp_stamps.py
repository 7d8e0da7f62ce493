"""In-kernel phase timing of icp_fit_kernel (diagnostic; SLAM_ICP_STAMPS=1): s_memtime ticks per
wavefront and iteration in [search, reduce, barrier wait, solve], sweep queries that fell back
to the ring search, and the search time of iterations 0-3."""
import os, sys, ctypes as C
os.environ["SLAM_ICP_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map(); batch = synth.make_batch(256)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for lanes in (0, -2, 2):
    icp = api.Icp(m_ga, m_nga, max_iter=iters, min_delta=-1.0, lanes_per_point=lanes)
    icp.fit_batch(batch)
    out = (C.c_double * 9)()
    L = api.lib(); L.slam_icp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    api.check(L.slam_icp_debug_stamps(icp.h, out))
    v = np.array(out[:])
    late = max(iters - 6, 1)
    print("lanes=%d per iteration per wave: search %.0f reduce %.0f barrier %.0f solve %.0f | search of iteration 0: %.0f"
          " | iterations >= 6: search %.0f of which cooperative rounds %.0f, %.2f undecided queries/wave"
          % ((lanes,) + tuple(v[:4] / iters) + (v[5], v[6] / late, v[7] / late, v[4] / late)))
