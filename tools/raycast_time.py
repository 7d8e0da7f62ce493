"""Event-timed raycast of config 2's batch (true poses) for a sweep of grid parameters:
python tools/raycast_time.py wg_per_cu=1,2 seg=0,24"""
import itertools
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

sweep = {"wg_per_cu": [0], "seg": [0], "impl": [api.RAYCAST_TILED]}
for a in sys.argv[1:]:
    k, v = a.split("=")
    sweep[k] = [int(x) for x in v.split(",")]
S, GRID = int(sweep.pop('scans', [256])[0]), int(sweep.pop('grid', [2000])[0])
batch = synth.make_batch(S)
R = np.stack([np.array([[np.cos(p[2]), -np.sin(p[2])], [np.sin(p[2]), np.cos(p[2])]]) for p in batch.true_poses])
t = batch.true_poses[:, :2].copy()
d = [api.DeviceArray.from_host(a, dt) for a, dt in ((batch.pts, np.float64), (batch.scan_off, np.int32), (R, np.float64), (t, np.float64))]
st = api.Stream()
for wg, seg, impl in itertools.product(sweep["wg_per_cu"], sweep["seg"], sweep["impl"]):
    g = api.Grid(GRID, GRID, 0.05, rolling=0, min_cluster_points=20, raycast_impl=impl, raycast_wg_per_cu=wg, raycast_seg_items=seg)
    ev = [api.Event() for _ in range(42)]
    for k in range(41):
        ev[k].record(st)
        g.raycast_scans_dev(d[0], d[1], S, batch.n_points, d[2], d[3], st)
    ev[41].record(st)
    st.synchronize()
    ms = np.array([ev[k].elapsed_ms(ev[k + 1]) for k in range(1, 41)])
    print("wg_per_cu=%d seg=%d impl=%d: raycast call %.4f ms (min %.4f) %s" % (wg, seg, impl, ms.mean(), ms.min(), g.raycast_stats()))
    g.close()
