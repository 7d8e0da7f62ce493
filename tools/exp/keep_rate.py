#!/usr/bin/env python3
"""How often would a list-form query keep its neighbour from one iteration to the next?  (a CPU study on the oracle's pose traces:
no GPU, nothing of the product)  For a few config-2 scans: per iteration, per query, the nearest model point (per class), the gap to
the second nearest (capped at the halo radius 0.136 m), and the displacement bound ||dR||_F |p| + |dt|; a query 'keeps' while the
accumulated 2 * displacement stays below the gap measured at its last search."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle_lib as O
from slam_amd import synth

HALO = 0.136
m_ga, m_nga = synth.make_map()
mg = m_ga.astype(np.float32).astype(np.float64)
mn = m_nga.astype(np.float32).astype(np.float64)
batch = synth.make_batch(8, n_loop=256, first=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
model = O.IcpModel(m_ga, m_nga)
p = O.icp_params(max_iter=30, min_delta=-1.0, indist=5.0)
tot = np.zeros((30, 6))
for s in range(8):
    t_ga, t_nga = batch.scan(s)
    R0, t0 = batch.R[s].reshape(2, 2), batch.t[s]
    _, _, trace, steps = model.fit(t_ga, t_nga, R0.copy(), t0.copy(), p)
    poses = [(R0, t0)] + [(trace[i, :4].reshape(2, 2), trace[i, 4:6]) for i in range(steps)]
    slack = None
    for it in range(30):
        R, t = poses[it]
        gaps, pn, qs, nns = [], [], [], []
        for pts, mod in ((t_ga, mg), (t_nga, mn)):
            if len(pts) == 0 or len(mod) <= 3:
                continue
            q = (pts @ R.T + t).astype(np.float32).astype(np.float64)
            d = np.sqrt(((q[:, None, :] - mod[None, :, :]) ** 2).sum(-1))
            part = np.partition(d, 1, axis=1)
            d1, d2 = part[:, 0], np.minimum(part[:, 1], HALO)
            g = np.where(d1 < HALO, d2 - d1, -1.0)     # not certified by the lists: no cache
            gaps.append(g); pn.append(np.abs(pts).sum(1)); qs.append(q); nns.append(np.argmin(d, axis=1))
        gap = np.concatenate(gaps); pnorm = np.concatenate(pn); qq = np.concatenate(qs); nn = np.concatenate(nns)
        if it > 0:
            Rp, tp = poses[it - 1]
            move = np.linalg.norm(R - Rp) * pnorm + np.linalg.norm(t - tp) + 8e-6
            slack = slack - move
        if slack is None:
            keep = np.zeros(len(gap), bool)
        else:
            keep = slack > 0
        slack = np.where(keep, slack, gap / 2 - 1e-6)
        if it > 0:
            slack2 = slack2 - (np.linalg.norm(qq - q_prev, axis=1) + 1e-6)
            keep2 = slack2 > 0
            same = (nn == nn_prev).sum()
        else:
            keep2 = np.zeros(len(gap), bool); same = 0
        slack2 = np.where(keep2, slack2 if it > 0 else 0, gap / 2 - 1e-6)
        q_prev, nn_prev = qq, nn
        tot[it] += [keep.sum(), len(gap), (gap < 0).sum(), keep2.sum(), same, trace[min(it, steps - 1), 6]]
for it in range(30):
    print("iter %2d: keep(bound) %.3f  keep(exact displacement) %.3f  nn unchanged %.3f  uncertified %.3f  mean delta %.2e" % (it, tot[it, 0] / tot[it, 1], tot[it, 3] / tot[it, 1], tot[it, 4] / tot[it, 1], tot[it, 2] / tot[it, 1], tot[it, 5] / 8))
