"""The model TILE of a wavefront (slam_icp_params::wave_tiles, icp_search.hpp): a model whose index does not fit LDS -- the
reference's own cap is 2 x 19 999 points (icpTools.h:21 ICP_MAX_PTS, icpTools.cpp:255-274, icp.cpp:51-60: both classes are copied) --
is searched out of per-wavefront LDS tiles staged from the index in HBM/L2.  It must find the SAME neighbours as the path it
replaces: the same iteration and correspondence counts as the untiled kernel, poses equal to rounding (the beams are dealt to the
wavefronts in another order), and the oracle's kd-tree fit at full size (config 2's 256 scans x 30 iterations, against 2 x 19 999 model points)."""
import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth
from test_gpu_icp import POS_TOL, ANG_TOL, yaw, ang_diff

pytestmark = pytest.mark.gpu


def model_at_the_cap(kind):
    """2 x 19 999 points: `room` = the synthetic room's walls (class NGA) and pillars (class GA) sampled 19 999 times each;
    `uniform` = 19 999 points per class uniformly random over the room (nothing to register against: the searches still must agree)."""
    if kind == "uniform":
        rs = np.random.RandomState(99)
        box = np.array([synth.ROOM_W, synth.ROOM_H])
        return rs.rand(19999, 2) * box - box / 2, rs.rand(19999, 2) * box - box / 2
    ga = synth.make_map(200000, seed=7)[0][:19999]
    nga = synth.make_map(60000, seed=8)[1][:19999]
    assert len(ga) == 19999 and len(nga) == 19999
    return np.ascontiguousarray(ga), np.ascontiguousarray(nga)


@pytest.mark.parametrize("kind", ["room", "uniform"])
def test_full_size_batch_at_the_model_cap_matches_oracle(kind):
    m_ga, m_nga = model_at_the_cap(kind)
    batch = synth.make_batch(256, n_loop=256)
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(30, -1.0, 5.0), n_threads=16)
    out = {}
    for tiles in (1, 0):
        icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, wave_tiles=tiles)
        assert not icp.index_info()["in_lds"]
        R, t, res, _ = icp.fit_batch(batch, indist=5.0)
        assert np.array_equal(res["iters"], iters) and (iters == 30).all()
        assert np.array_equal(res["n_corr"], ncorr)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
        out[tiles] = (R, t, res["delta"].copy())
        icp.close()
    # tiled and untiled: the same neighbour for every query; the tiled form deals the beams to the wavefronts in another order
    # (a wavefront's two passes are neighbours), so the sums agree to rounding, not to the bit
    for a, b in zip(out[1], out[0]):
        assert np.abs(a - b).max() < 1e-9


@pytest.mark.parametrize("mode", ["p2p", "p2l"])
def test_tiles_on_a_small_model_kept_out_of_lds(mode):
    """force_global puts config 2's own 10 k-point map in HBM/L2: the tiled ring form against the untiled one (bitwise), early exit
    on min_delta included, ragged scans, and the point-to-line step (one class, a normal per neighbour)."""
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(24, n_loop=256)
    # ragged: cut some scans short, one below a wavefront's worth, one of 5 points (the minimum, icp.cpp:100-103)
    keep = [None] * batch.n_scans
    keep[3], keep[7], keep[11] = 700, 40, 5
    parts, nga, off = [], [], [0]
    for s in range(batch.n_scans):
        p = batch.pts[batch.scan_off[s]:batch.scan_off[s + 1]]
        n = keep[s] or len(p)
        parts.append(p[:n]); nga.append(min(int(batch.scan_nga[s]), n)); off.append(off[-1] + n)
    rag = synth.ScanBatch(np.ascontiguousarray(np.concatenate(parts)), np.array(off, np.int32), np.array(nga, np.int32), batch.R, batch.t,
                          batch.true_poses)
    kw = dict(mode=api.ICP_P2L, normals_k=10) if mode == "p2l" else {}
    for max_iter, min_delta in ((30, -1.0), (60, 1e-6)):
        out = {}
        for tiles in (1, 0):
            icp = api.Icp(m_ga, m_nga, max_iter=max_iter, min_delta=min_delta, force_global=1, lanes_per_point=2, spread_scans=-1,
                          wave_tiles=tiles, **kw)
            assert not icp.index_info()["in_lds"]
            R, t, res, _ = icp.fit_batch(rag, indist=5.0)
            out[tiles] = (R, t, res["iters"].copy(), res["n_corr"].copy(), res["delta"].copy())
            icp.close()
        assert np.array_equal(out[1][2], out[0][2]) and np.array_equal(out[1][3], out[0][3])      # iterations, correspondences
        for a, b in zip(out[1], out[0]):
            assert np.abs(a - b).max() < 1e-9
        if mode == "p2p":
            model = O.IcpModel(m_ga, m_nga)
            Ro, to, iters, ncorr, delta = model.fit_batch(rag.pts, rag.scan_off, rag.scan_nga, rag.R, rag.t, O.icp_params(max_iter, min_delta, 5.0))
            assert np.array_equal(out[1][2], iters) and np.array_equal(out[1][3], ncorr)
            assert np.abs(out[1][1] - to).max() < POS_TOL and ang_diff(yaw(out[1][0]), yaw(Ro)).max() < ANG_TOL


def test_exact_ties_through_the_tiled_form():
    """The gridded, duplicated model of test_exact_ties_through_every_batch_form, kept out of LDS: a tie met inside a tile sends the
    query to the exact pass on the index in L2 (lowest original index, kdtree.cpp:360-375)."""
    from test_gpu_icp_spread import check_against_oracle
    rs = np.random.RandomState(23)
    gx, gy = np.meshgrid(np.arange(60) * 0.5, np.arange(40) * 0.5)
    grid = np.stack([gx.ravel(), gy.ravel()], 1)
    m_nga = np.concatenate([grid, grid[:600]])
    m_ga = np.concatenate([grid[1000:1400] + [0.25, 0.0], grid[1000:1100] + [0.25, 0.0]])
    scans, nga, Rs, ts = [], [], [], []
    for k in range(24):
        kind = k % 3
        pick = grid[rs.choice(len(grid), 400)]
        if kind == 0:
            pts, pose = pick + 0.25, (0.0, 0.0, 0.0)
        elif kind == 1:
            pts, pose = pick.copy(), (0.0, 0.0, 0.0)
        else:
            pts, pose = pick + rs.randn(400, 2) * 0.03, (0.05, -0.04, 0.004)
        R0, t0 = synth.pose_to_Rt(*pose)
        scans.append(pts); nga.append(60 if k % 2 else 0); Rs.append(R0.reshape(4)); ts.append(t0)
    off = np.cumsum([0] + [len(x) for x in scans]).astype(np.int32)
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(scans)), off, np.array(nga, np.int32), np.array(Rs), np.array(ts),
                            np.zeros((len(scans), 3)))
    icp, R, t, res, tr = check_against_oracle(m_ga, m_nga, batch, 12, 1e-9, nn=O.NN_BRUTE, spread_scans=-1, lanes_per_point=2,
                                              force_global=1, wave_tiles=1)
    assert not icp.index_info()["in_lds"] and (res["n_corr"] > 300).all()
    icp.close()
