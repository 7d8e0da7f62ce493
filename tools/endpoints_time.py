"""MLS::addToOccupancy's endpoint update (slam_grid_add_endpoints_dev) for a 64-ring cloud on local_mapper's grid
(200 x 200 @ 0.2 m, rolling) and on a 2000 x 2000 @ 0.05 m one: event-timed."""
import sys
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from slam_amd import api, synth
import oracle_lib as O

xyz = synth.make_cloud3d(3, n_loop=50)[0]
lab = O.gseg_segment(xyz)[0]
obs = np.concatenate([xyz[lab >= O.GSEG_OBSTACLE], np.zeros((int((lab >= O.GSEG_OBSTACLE).sum()), 1), np.float32)], 1)
gnd = np.concatenate([xyz[lab == O.GSEG_GROUND], np.zeros((int((lab == O.GSEG_GROUND).sum()), 1), np.float32)], 1)
d_obs, d_gnd = api.DeviceArray.from_host(obs), api.DeviceArray.from_host(gnd)
L = api.lib()
st = api.Stream()
for size, res, rolling in ((200, 0.2, 1), (2000, 0.05, 0)):
    g = api.Grid(size, size, res, rolling=rolling, min_cluster_points=20)
    ev = [api.Event() for _ in range(22)]
    for k in range(21):
        ev[k].record(st)
        api.check(L.slam_grid_add_endpoints_dev(g.h, d_obs.ptr, len(obs), d_gnd.ptr, len(gnd), 4, st.ptr))
    ev[21].record(st)
    st.synchronize()
    ms = np.array([ev[k].elapsed_ms(ev[k + 1]) for k in range(1, 21)])
    print("%d x %d @ %.2f: %d obstacle + %d ground points: %.4f ms (min %.4f), %d updates per call" %
          (size, size, res, len(obs), len(gnd), ms.mean(), ms.min(), g.total_updates() // 21))
    g.close()
