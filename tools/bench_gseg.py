#!/usr/bin/env python3
"""Times the ground-segmentation pre-filter on a config-3
sized cloud (64 rings x 2048 azimuth steps) with HIP events, and the CPU oracle
beside it.  Prints one JSON line (not the headline bench)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from slam_amd import api, synth
import oracle_lib as O

xyz = synth.make_cloud3d(3, n_loop=50)[0]
n = len(xyz)
seg = api.GroundSegmentation()
d_xyz = api.DeviceArray.from_host(xyz)
d_lab = api.DeviceArray((n,), np.uint8)
d_gnd = api.DeviceArray((n, 4), np.float32)
d_obs = api.DeviceArray((n, 4), np.float32)
d_cnt = api.DeviceArray((2,), np.int32)
for _ in range(3):
    seg.segment_dev(d_xyz, n, 3, d_lab)
    seg.split_dev(d_xyz, n, 3, d_lab, d_gnd, d_obs, d_cnt)
api.synchronize()
K = 20
e = [api.Event() for _ in range(3)]
ms_seg = ms_split = 0.0
for _ in range(K):
    e[0].record(); seg.segment_dev(d_xyz, n, 3, d_lab); e[1].record()
    seg.split_dev(d_xyz, n, 3, d_lab, d_gnd, d_obs, d_cnt); e[2].record()
    e[2].synchronize()
    ms_seg += e[0].elapsed_ms(e[1]); ms_split += e[1].elapsed_ms(e[2])
t0 = time.perf_counter()
for _ in range(5):
    O.gseg_segment(xyz)
cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"workload": "ground segmentation, %d points (64 x 2048 rays)" % n,
                  "gpu_segment_ms": ms_seg / K, "gpu_split_ms": ms_split / K,
                  "gpu_points_per_s": n / (ms_seg / K * 1e-3), "cpu_oracle_1thread_ms": cpu_ms,
                  "insac_iterations": int(seg.read_model()[2].sum())}))
