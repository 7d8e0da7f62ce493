#!/usr/bin/env python3
"""Generates tests/golden/spread_case3.npz: three matches of BASELINE config 3 as the ICP sees them -- the target cloud 0 of
synth.make_cloud3d through the ORACLE's chain (ground segmentation, GA/NGA classification, bin order, crop + split with the
19 999 cap: icpTools.cpp:36-103, 225-276) = 19 999 GA + 871 NGA model points, wall points of 64 rings stacked in 2-D (cells of
hundreds of points); scenes = clouds 1, 5, 9 through the chain with the voxel filter (~600 points each, in the filter's order,
which has no locality); initial poses as tools/bench_config3.py sets them.  Inputs only: the expected poses are computed by
the oracle inside the tests.  Model coordinates are float32 values (as the chain leaves them), scenes are the centroids' doubles."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import oracle_lib as O
from slam_amd import synth


def obstacle(B):
    lab, *_ = O.gseg_segment(B)
    return B[lab >= O.GSEG_OBSTACLE]


A, pa = synth.make_cloud3d(0, n_loop=50)
out_a = obstacle(A)
fa = O.classify_ga(out_a)
kept = np.flatnonzero(fa != 255)
bx = np.floor((out_a[:, 0].astype(np.float64) + 300) / 0.5).astype(np.int64)
by = np.floor((out_a[:, 1].astype(np.float64) + 300) / 0.5).astype(np.int64)
order = kept[np.argsort((bx * 1200 + by)[kept], kind="stable")]
seg_t = np.concatenate([out_a[order], (fa[order] == 1).astype(np.float32)[:, None]], 1)
m_ga, m_nga = O.ccicp_split(seg_t, O.ccicp_crop(seg_t, 0.0, 0.0))
out = {"m_ga": m_ga.astype(np.float32), "m_nga": m_nga.astype(np.float32)}
assert np.array_equal(out["m_ga"].astype(np.float64), m_ga) and np.array_equal(out["m_nga"].astype(np.float64), m_nga)
for k in (1, 5, 9):
    B, pb = synth.make_cloud3d(k, n_loop=50)
    out_b = obstacle(B)
    fb = O.classify_ga(out_b)
    kb = fb != 255
    seg_s, nv = O.voxel_downsample(np.concatenate([out_b[kb], fb[kb, None].astype(np.float32)], 1))
    s_ga, s_nga = O.ccicp_split(seg_s, None)
    ca, sa = np.cos(pa[2]), np.sin(pa[2])
    rel = (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])
    R0, t0 = synth.pose_to_Rt(rel[0] + 0.1, rel[1] - 0.1, rel[2] + 0.02)
    out["s_ga%d" % k], out["s_nga%d" % k], out["R%d" % k], out["t%d" % k] = s_ga, s_nga, R0, t0
np.savez_compressed(os.path.join(HERE, "spread_case3.npz"), **out)
print({k: v.shape for k, v in out.items()})
