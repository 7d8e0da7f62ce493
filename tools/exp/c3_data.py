"""Round 5: config 3's sequence as files for tests/cpp/ccicp_sequence (what tools/bench_config3.measure_cpp hands it), kept on disk so
that the program can be run by itself (under rocprofv3, with other forms):  python tools/exp/c3_data.py DIR [n_clouds=50] [advance=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slam_amd import synth
d, n, advance = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 50, int(sys.argv[3]) if len(sys.argv) > 3 else 10
os.makedirs(d, exist_ok=True)
clouds, poses = zip(*[synth.make_cloud3d(k, n_loop=50) for k in range(n)])
init, truth = [], []
for k in range(1, n):
    j = ((k - 1) // advance) * advance if advance > 0 else 0
    pa, pb = poses[j], poses[k]
    ca, sa = np.cos(pa[2]), np.sin(pa[2])
    rel = (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])
    yaw = rel[2] + 0.02
    init.append([rel[0] + 0.1, rel[1] - 0.1, 0.0, 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
    truth.append(list(rel))
for k, c in enumerate(clouds):
    np.ascontiguousarray(c, np.float32).tofile(os.path.join(d, "cloud%d.f32" % k))
np.array(init, np.float64).tofile(os.path.join(d, "init.f64"))
np.array(truth, np.float64).tofile(os.path.join(d, "truth.f64"))
