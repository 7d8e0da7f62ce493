"""How evenly the scans of a batch finish (diagnostic; SLAM_ICP_STAMPS=1): the kernel takes as long as its
slowest workgroup, one scan each.  Prints the distribution of per-scan time (s_memtime ticks of wavefront 0,
all phases, all iterations) and what the slowest scans have in common."""
import os, sys, ctypes as C
os.environ["SLAM_ICP_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map(); batch = synth.make_batch(256)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
icp = api.Icp(m_ga, m_nga, max_iter=iters, min_delta=-1.0, lanes_per_point=0)
res = icp.fit_batch(batch)
L = api.lib()
L.slam_icp_debug_stamps_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
S = batch.n_scans
raw = np.zeros((S * 16, 9), dtype=np.int64); rows = C.c_int(0)
api.check(L.slam_icp_debug_stamps_raw(icp.h, raw.ctypes.data, S * 16, C.byref(rows)))
raw = raw[:rows.value].reshape(-1, 16, 9)
tot = raw[:, 0, :4].sum(axis=1).astype(float)        # wavefront 0: search + reduce + wait + solve
n = np.diff(batch.scan_off)
order = np.argsort(-tot)
print("per-scan ticks: mean %.0f  median %.0f  p90 %.0f  max %.0f  (max/mean %.2f)" %
      (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), tot.max() / tot.mean()))
print("slowest scans: scan, ticks, points, search it0, search late, coop late, undecided late (wave mean)")
for s in list(order[:8]) + list(order[-3:]):
    w = raw[s].astype(float)
    print("  %3d %7.0f %5d | it0 %6.0f late %6.0f coop %6.0f undecided %5.1f | search %6.0f reduce %5.0f wait %6.0f solve %5.0f" %
          (s, tot[s], n[s], w[:, 5].mean(), w[:, 6].mean(), w[:, 7].mean(), w[:, 4].mean(),
           w[:, 0].mean(), w[:, 1].mean(), w[:, 2].mean(), w[:, 3].mean()))
print("correlation of time with points: %.2f" % np.corrcoef(tot, n)[0, 1])
for s in (order[0], order[-1]):
    w = raw[s].astype(float)
    print("scan %d per wavefront: search (all iterations) / barrier wait / search it0 / search late / coop late" % s)
    for k in range(16):
        print("   wave %2d: %7.0f %7.0f %6.0f %7.0f %6.0f" % (k, w[k, 0], w[k, 2], w[k, 5], w[k, 6], w[k, 7]))
