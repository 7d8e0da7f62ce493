#!/usr/bin/env python3
"""BASELINE config 3 chain on one cloud (64 rings x 2048 azimuth steps, ~131 k rays), device-resident
through the C-ABI: ground segmentation -> obstacle/ground split -> GA/NGA classification -> voxel filter ->
crop + class split (-> the arrays slam_icp_create takes) -> height recovery; wall-clock per stage with the
CPU oracle beside it.  Prints one JSON line (not the headline bench)."""
import ctypes as C
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from slam_amd import api, synth
import oracle_lib as O

xyz = synth.make_cloud3d(3, n_loop=50)[0]
n = len(xyz)
L = api.lib()
seg, cc = api.GroundSegmentation(), api.Ccicp()
d_xyz = api.DeviceArray.from_host(xyz)
d_lab = api.DeviceArray((n,), np.uint8)
d_gnd = api.DeviceArray((n, 4), np.float32)
d_obs = api.DeviceArray((n, 4), np.float32)
d_cnt = api.DeviceArray((2,), np.int32)
d_cnt2 = api.DeviceArray((1,), np.int32)
d_drv = api.DeviceArray((n, 4), np.float32)
d_flag = api.DeviceArray((n,), np.uint8)
d_vox = api.DeviceArray((n, 4), np.float32)
d_ga = api.DeviceArray((20000, 2), np.float64)
d_nga = api.DeviceArray((20000, 2), np.float64)
pose = (C.c_double * 7)(5.0, 0.0, 0.0, 0, 0, 0, 1)


def chain(timing=None):
    t = [time.perf_counter()]
    seg.segment_dev(d_xyz, n, 3, d_lab)
    seg.split_dev(d_xyz, n, 3, d_lab, d_gnd, d_drv, d_cnt)               # ground cloud (+ drvCloud for the grid)
    c_obs = C.c_int(0)
    api.check(L.slam_ccicp_select_dev(cc.h, d_xyz.ptr, n, 3, d_lab.ptr, (1 << 2) | (1 << 3), d_obs.ptr, C.byref(c_obs), None))
    t.append(time.perf_counter())
    n_gnd = int(d_cnt.download()[0]); n_obs = c_obs.value
    api.check(L.slam_gseg_classify_ga_dev(seg.h, d_obs.ptr, n_obs, 4, d_flag.ptr, None))
    api.synchronize(); t.append(time.perf_counter())
    n_vox = C.c_int(0)
    api.check(L.slam_ccicp_voxel_downsample_dev(cc.h, d_obs.ptr, d_flag.ptr, n_obs, 4, 0.5, 0.5, 2.0, d_vox.ptr, n,
                                                C.byref(n_vox), None))
    t.append(time.perf_counter())
    counts = (C.c_int * 2)()
    api.check(L.slam_ccicp_split_dev(cc.h, d_vox.ptr, n_vox.value, 4, 1, 0.0, 0.0, 75.0, 20000, d_ga.ptr, d_nga.ptr,
                                     counts, None))
    t.append(time.perf_counter())
    z, nc = C.c_double(0), C.c_int(0)
    api.check(L.slam_ccicp_height_dev(cc.h, d_gnd.ptr, n_gnd, 4, pose, C.byref(z), C.byref(nc), None, None))
    t.append(time.perf_counter())
    if timing is not None:
        timing.append(np.diff(t) * 1e3)
    return n_obs, n_gnd, n_vox.value, counts[0], counts[1], z.value, nc.value


for _ in range(3):
    out = chain()
times = []
for _ in range(20):
    out = chain(times)
ms = np.mean(times, axis=0)

t0 = time.perf_counter()
lab, *_ = O.gseg_segment(xyz)
t1 = time.perf_counter()
obs = xyz[lab >= O.GSEG_OBSTACLE]
flags = O.classify_ga(obs)
t2 = time.perf_counter()
keep = flags != 255
vox, nv = O.voxel_downsample(np.concatenate([obs[keep], flags[keep, None].astype(np.float32)], 1))
t3 = time.perf_counter()
ga, nga = O.ccicp_split(vox, O.ccicp_crop(vox, 0.0, 0.0))
t4 = time.perf_counter()
zo, nco, _ = O.ccicp_height(xyz[lab == O.GSEG_GROUND], list(pose))
t5 = time.perf_counter()
print(json.dumps({
    "workload": "config 3 front end, one %d-point cloud" % n,
    "stages": ["segment+split+select", "classify GA", "voxel 0.5/0.5/2", "crop+split", "height"],
    "gpu_ms": [round(float(v), 4) for v in ms], "gpu_total_ms": round(float(ms.sum()), 4),
    "cpu_oracle_1thread_ms": [round(v * 1e3, 3) for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)],
    "obstacle_points": out[0], "ground_points": out[1], "voxels": out[2], "model_ga": out[3], "model_nga": out[4],
    "z": out[5], "z_oracle": zo, "counts_match_oracle": bool(out[2] == nv and out[3] == len(ga) and out[4] == len(nga)),
    "clouds_per_s": 1e3 / float(ms.sum()), "points_per_s": n * 1e3 / float(ms.sum())}))
