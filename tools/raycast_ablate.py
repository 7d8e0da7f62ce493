"""Where the tiled raycast's time goes: the raycast call of config 2's batch (true poses) event-timed with parts of the
kernel switched off (measurement build, SLAM_RAYCAST_ABLATE bits: 1 = no walk, 2 = no write-back, 8 = no blocks at all,
16 = no clipping and no walk) -- counts are wrong then, only the time is of interest.
    python -m slam_amd.build --measure && SLAM_AMD_MEASURE=1 python tools/raycast_ablate.py [wg=2] [seg=32] [scans=256] [grid=2000]"""
import os
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np

assert os.environ.get("SLAM_AMD_MEASURE") == "1", "needs the measurement build: SLAM_AMD_MEASURE=1"
from slam_amd import api, synth

opt = dict(wg=0, seg=0, scans=256, grid=2000)
for a in sys.argv[1:]:
    k, v = a.split("=")
    opt[k] = int(v)
batch = synth.make_batch(opt["scans"])
R = np.stack([np.array([[np.cos(p[2]), -np.sin(p[2])], [np.sin(p[2]), np.cos(p[2])]]) for p in batch.true_poses])
t = batch.true_poses[:, :2].copy()
d = [api.DeviceArray.from_host(a, dt) for a, dt in ((batch.pts, np.float64), (batch.scan_off, np.int32), (R, np.float64), (t, np.float64))]
st = api.Stream()
for abl, what in ((0, "everything"), (2, "no write-back"), (1, "no walk"), (3, "no walk, no write-back"), (17, "no clipping, no walk"),
                  (19, "no clipping, no walk, no write-back"), (8, "no blocks (pre-pass + work list + launch)")):
    os.environ["SLAM_RAYCAST_ABLATE"] = str(abl)
    g = api.Grid(opt["grid"], opt["grid"], 0.05, rolling=0, min_cluster_points=20, raycast_wg_per_cu=opt["wg"], raycast_seg_items=opt["seg"])
    ev = [api.Event() for _ in range(32)]
    for k in range(31):
        ev[k].record(st)
        g.raycast_scans_dev(d[0], d[1], opt["scans"], batch.n_points, d[2], d[3], st)
    ev[31].record(st)
    st.synchronize()
    ms = np.array([ev[k].elapsed_ms(ev[k + 1]) for k in range(1, 31)])
    print("ablate=%2d %-45s raycast call %.4f ms (min %.4f) %s" % (abl, what, ms.mean(), ms.min(), g.raycast_stats()), flush=True)
    g.close()
