"""Randomised models through the device-planned index build (icp_build.hip: plan kernels on the device, one host wait) against
the single-threaded host build: the same plan (index_info) and the same bytes, cell index and halo lists.  A soak to run by
hand after touching the build.    timeout -k 10 400 python tools/soak_build.py [seconds]"""
import sys
import time

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(11)
wall_ga, wall_nga = synth.make_map(20000)
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    kind = rs.randint(0, 6)
    n_ga, n_nga = int(rs.randint(0, 9000)), int(rs.randint(5, 12000))
    scale = 10.0 ** rs.uniform(-2, 2.5)
    off = rs.uniform(-1, 1, 2) * 10.0 ** rs.uniform(-1, 3)
    if kind == 0:      # the synthetic room, a random part of it
        ga, nga = wall_ga[rs.permutation(len(wall_ga))[:n_ga]], wall_nga[rs.permutation(len(wall_nga))[:n_nga]]
    elif kind == 1:    # gaussian blobs
        ga, nga = rs.randn(n_ga, 2) * scale + off, rs.randn(n_nga, 2) * scale * rs.uniform(0.1, 3) + off
    elif kind == 2:    # uniform boxes of odd aspect
        ga = rs.rand(n_ga, 2) * [scale, scale * 10.0 ** rs.uniform(-3, 0)] + off
        nga = rs.rand(n_nga, 2) * [scale * 10.0 ** rs.uniform(-3, 0), scale] + off
    elif kind == 3:    # lattice with duplicates
        gx, gy = np.meshgrid(np.arange(60) * scale * 0.01, np.arange(50) * scale * 0.01)
        grid = np.stack([gx.ravel(), gy.ravel()], 1) + off
        ga, nga = grid[rs.randint(0, len(grid), n_ga)], grid[rs.randint(0, len(grid), n_nga)]
    elif kind == 4:    # one dense wall plus clutter
        w = np.stack([rs.rand(n_nga) * scale, off[1] + rs.randn(n_nga) * 0.002], 1)
        ga, nga = rs.rand(n_ga, 2) * scale + off, w
    else:              # tiny models
        ga, nga = rs.rand(int(rs.randint(0, 4)), 2) * scale, rs.rand(int(rs.randint(5, 40)), 2) * scale + off
    if rs.rand() < 0.15 and len(nga) > 50:   # non-finite points
        nga = nga.copy()
        nga[rs.randint(0, len(nga), 5)] = [np.nan, np.inf]
    kw = {}
    r = rs.rand()
    if r < 0.15: kw["force_global"] = 1
    elif r < 0.3: kw["cell_size"] = float(10.0 ** rs.uniform(-1.5, 0.5) * max(scale, 0.05))
    elif r < 0.4: kw["lanes_per_point"] = 8
    dev = api.Icp(ga, nga, **kw)
    host = api.Icp(ga, nga, build_on_host=1, **kw)
    a, b = dev.index_info(), host.index_info()
    assert a == b, (kind, len(ga), len(nga), kw, a, b)
    for which in (0, 1):
        x, y = dev.index_blob(which), host.index_blob(which)
        assert x.shape == y.shape and np.array_equal(x, y), (kind, len(ga), len(nga), kw, which, x.shape, y.shape)
    dev.close(); host.close()
    n += 1
print("soak ok: %d models, device build == host build, in %.0f s" % (n, budget))
