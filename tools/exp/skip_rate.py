"""Offline (CPU, numpy/scipy): how often could a list-form query of iteration k keep its neighbour of iteration k-1 on a
triangle-inequality certificate?  For each scan of config 2: run P2P ICP (float64 here: a rate estimate, not parity), and per
iteration count points with  sqrt(d_new(m)) + move < sqrt(d2_ref) - move', where d2_ref is the SECOND neighbour's distance at the
last full search of that point and move the distance the query has travelled since."""
import sys, os
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from slam_amd import synth

m_ga, m_nga = synth.make_map(10000)
model = np.concatenate([m_ga, m_nga])[:, :2] if len(m_ga) else m_nga[:, :2]
trees = [cKDTree(m_ga[:, :2]) if len(m_ga) > 3 else None, cKDTree(m_nga[:, :2])]
mods = [m_ga[:, :2], m_nga[:, :2]]
batch = synth.make_batch(16)
tot = np.zeros((30, 3))
for s in range(batch.n_scans):
    a, e = batch.scan_off[s], batch.scan_off[s + 1]
    P = batch.pts[a:e, :2].astype(np.float64)
    nga = batch.scan_nga[s]
    cls = (np.arange(e - a) >= nga).astype(int)
    R = batch.R[s].reshape(2, 2).copy(); t = batch.t[s].copy()
    ref_q = None; ref_d2 = None; ref_nn = None
    for it in range(30):
        q = P @ R.T + t
        nn = np.zeros(len(P), int); d1 = np.zeros(len(P)); d2 = np.zeros(len(P))
        for c in (0, 1):
            sel = cls == c
            if trees[c] is None or not sel.any():
                continue
            d, i = trees[c].query(q[sel], k=2)
            nn[sel] = i[:, 0]; d1[sel] = d[:, 0]; d2[sel] = d[:, 1]
        if ref_q is not None:
            move = np.linalg.norm(q - ref_q, axis=1)
            mcoord = np.where(cls[:, None] == 0, mods[0][np.minimum(ref_nn, len(mods[0]) - 1)], mods[1][np.minimum(ref_nn, len(mods[1]) - 1)])
            dn = np.linalg.norm(q - mcoord, axis=1)
            ok = dn + move < ref_d2 * (1 - 1e-5) - 1e-7
            tot[it, 0] += ok.sum(); tot[it, 1] += len(P); tot[it, 2] += np.mean(move)
            # points whose certificate failed are searched in full: new reference
            ref_q = np.where(ok[:, None], ref_q, q); ref_d2 = np.where(ok, ref_d2, d2); ref_nn = np.where(ok, ref_nn, nn)
            assert (ref_nn == nn)[ok].all()
        else:
            ref_q, ref_d2, ref_nn = q.copy(), d2.copy(), nn.copy()
        inl = d1 < 5.0
        A = q[inl]; B = np.where(cls[inl, None] == 0, mods[0][np.minimum(nn[inl], len(mods[0]) - 1)], mods[1][nn[inl]])
        ca, cb = A.mean(0), B.mean(0)
        H = (A - ca).T @ (B - cb)
        U, S, Vt = np.linalg.svd(H)
        dR = Vt.T @ U.T
        dt = cb - dR @ ca
        R = dR @ R; t = dR @ t + dt
for it in range(1, 30):
    print(it, "skip %.3f" % (tot[it, 0] / tot[it, 1]), "mean move %.2e" % (tot[it, 2] / batch.n_scans))
