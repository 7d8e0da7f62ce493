#!/usr/bin/env python3
"""Point-to-line forms against the oracle, step by step: which scans differ in the first step and in what."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from slam_amd import api, synth

m_ga, m_nga = synth.make_map(5000)
model = O.IcpModel(m_ga, m_nga, normals_k=10)
prm = O.icp_params(15, -1.0, 5.0, O.NN_KDTREE, O.MODE_P2L)
batch = synth.make_batch(40, n_loop=256)
allm = np.concatenate([m_ga, m_nga]).astype(np.float32)
for name, kw in (("ring-only", dict(spread_scans=-1, lanes_per_point=2)), ("fused", dict(spread_scans=-1)), ("spread", dict())):
    icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10, max_iter=15, min_delta=-1.0, **kw)
    nrm_g, nrm_o = icp.normals(), model.normals()
    print(name, icp.index_info(), "normal dot min", np.abs((nrm_g * nrm_o).sum(1)).min(), "max comp diff up to sign",
          np.minimum(np.abs(nrm_g - nrm_o).max(1), np.abs(nrm_g + nrm_o).max(1)).max())
    R, t, res, trace = icp.fit_batch(batch, trace=True)
    for s in range(batch.n_scans):
        t_ga, t_nga = batch.scan(s)
        Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], prm)
        d0 = abs(trace[s, 0, 6] - tr[0, 6])
        if d0 > 1e-9:
            d, R1, t1, nc, corr = model.fit_step(t_ga, t_nga, batch.R[s], batch.t[s], prm)
            q = np.concatenate([O.transform_points(t_ga, batch.R[s], batch.t[s]), O.transform_points(t_nga, batch.R[s], batch.t[s])])
            dis, idx = icp.nearest(1, q)
            bad = np.nonzero(idx != corr)[0]
            print("  scan %d: delta0 diff %.3e, %d queries with another neighbour" % (s, d0, len(bad)))
            for i in bad[:5]:
                dg = ((allm[idx[i]] - q[i]).astype(np.float32) ** 2).sum(dtype=np.float32)
                do = ((allm[corr[i]] - q[i]).astype(np.float32) ** 2).sum(dtype=np.float32)
                print("    q", q[i], "gpu", idx[i], dis[i], dg, "oracle", corr[i], do)
    icp.close()
