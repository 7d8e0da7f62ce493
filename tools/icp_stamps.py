import os, sys, ctypes as C
os.environ["SLAM_ICP_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map(); batch = synth.make_batch(256)
for lanes in (0, 1, 2, 4):
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, lanes_per_point=lanes)
    icp.fit_batch(batch)
    out = (C.c_double * 4)()
    L = api.lib(); L.slam_icp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    api.check(L.slam_icp_debug_stamps(icp.h, out))
    v = np.array(out[:]) / 30.0
    print("lanes=%d cycles per iteration per wave (100 MHz ticks x?): search %.0f reduce %.0f barrier %.0f solve %.0f" % (lanes, *v))
