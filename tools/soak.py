"""A few minutes of randomised launches of the pieces with spin-waits in them (the raycast's ticket protocol, the spread form's
exchange with its hand-over), each checked against the forms without them.  Not a test of the suite: a soak to run by hand
under a timeout after touching those kernels.    timeout -k 10 400 python tools/soak.py [seconds]"""
import sys
import time

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(7)
m_ga, m_nga = synth.make_map(10000)
t_end = time.time() + budget
n_ray = n_fit = 0
while time.time() < t_end:
    # ---- raycast: random batch, grid, chunk, workgroups; tiled == global atomics
    S = int(rs.randint(1, 200))
    size = int(rs.choice([300, 700, 1500, 2000, 3000]))
    res = float(rs.choice([0.05, 0.1, 0.2]))
    batch = synth.make_batch(S, n_loop=256, first=int(rs.randint(0, 200)))
    Rt = [synth.pose_to_Rt(*p) for p in batch.true_poses]
    R = np.stack([r.reshape(4) for r, _ in Rt]); t = np.stack([tt for _, tt in Rt])
    d = [api.DeviceArray.from_host(a, dt) for a, dt in ((batch.pts, np.float64), (batch.scan_off, np.int32), (R, np.float64), (t, np.float64))]
    ref = api.Grid(size, size, res, rolling=0, raycast_impl=api.RAYCAST_GLOBAL)
    ref.raycast_scans_dev(d[0], d[1], S, batch.n_points, d[2], d[3])
    api.synchronize()
    H, M = ref.read_counts(); ref.close()
    g = api.Grid(size, size, res, rolling=0, raycast_seg_items=int(rs.choice([0, 8, 13, 16, 40, 200, 511])),
                 raycast_wg_per_cu=int(rs.choice([0, 1, 2])), raycast_max_workgroups=int(rs.choice([0, 0, 1, 2, 7, 100])))
    for rep in (1, 2, 3):
        g.raycast_scans_dev(d[0], d[1], S, batch.n_points, d[2], d[3])
    api.synchronize()
    h, m = g.read_counts()
    assert np.array_equal(h, 3 * H) and np.array_equal(m, 3 * M), ("raycast", S, size, res)
    g.close(); n_ray += 1
    # ---- spread form: random small batch, sometimes handed over at once; == the workgroup-per-scan form to rounding
    S = int(rs.randint(1, 16))
    b = synth.make_batch(S, n_loop=256, first=int(rs.randint(0, 200)))
    a = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6, spread_wait_us=int(rs.choice([0, 0, -1, 1, 50])))
    c = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6, spread_scans=-1)
    Ra, ta, ra, _ = a.fit_batch(b); Rc, tc, rc, _ = c.fit_batch(b)
    assert np.array_equal(ra["iters"], rc["iters"]) and np.array_equal(ra["n_corr"], rc["n_corr"]), ("spread", S)
    assert np.abs(ta - tc).max() < 1e-9 and np.abs(Ra - Rc).max() < 1e-9
    a.close(); c.close(); n_fit += 1
print("soak ok: %d raycast cases, %d spread cases in %.0f s" % (n_ray, n_fit, budget))
