"""Where the INSAC kernel's time goes (measurement build: SLAM_AMD_MEASURE=1 python tools/exp/insac_time.py)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import api, synth
L = api.lib()
seg = api.GroundSegmentation()
B, _ = synth.make_cloud3d(1, n_loop=50)
d_xyz = api.DeviceArray.from_host(B, np.float32); d_lab = api.DeviceArray((len(B),), np.uint8)
for _ in range(3): seg.segment_dev(d_xyz, len(B), 3, d_lab)
api.synchronize()
a, b = api.Event(), api.Event(); a.record(); seg.segment_dev(d_xyz, len(B), 3, d_lab); b.record(); b.synchronize()
print("segment_dev %.1f us" % (a.elapsed_ms(b) * 1e3))
out = (C.c_double * 8)()
if hasattr(L, "slam_gseg_debug_insac") and L.slam_gseg_debug_insac(out) == 0:
    print("mean us per sector: setup %.2f matrix %.2f factorisation %.2f solves %.2f candidates %.2f verdict %.2f | rounds %.2f | slowest sector %.2f" % tuple(out))
