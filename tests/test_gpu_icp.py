"""GPU parity of the ICP path through the C-ABI against the CPU oracle
(icp.cpp / icpPointToPoint.cpp / kdtree.cpp restated) and the golden chain.

Tolerances (BASELINE.json north_star): pose within 1e-4 m / 1e-5 rad of the
reference arithmetic; correspondences (float NN distance and index) bit-exact.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth

pytestmark = pytest.mark.gpu

POS_TOL, ANG_TOL = 1e-4, 1e-5


def yaw(R):
    R = np.asarray(R).reshape(-1, 4)
    return np.arctan2(R[:, 2], R[:, 0])  # icpTools.cpp:197 atan2(R10, R00)


def ang_diff(a, b):
    d = a - b
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


@pytest.fixture(scope="module")
def world():
    m_ga, m_nga = synth.make_map()
    return m_ga, m_nga, O.IcpModel(m_ga, m_nga)


@pytest.mark.parametrize("lanes", [1, 4, 8, 16, 64])
@pytest.mark.parametrize("force_global", [0, 1])
def test_nearest_is_bit_exact(world, lanes, force_global):
    """KDTree::n_nearest(q,1) contract: same float distance, same index
    (kdtree.cpp:378-391; brute force of :360-375 as the tie-free arbiter)."""
    m_ga, m_nga, _ = world
    icp = api.Icp(m_ga, m_nga, lanes_per_point=lanes, force_global=force_global)
    info = icp.index_info()
    assert info["in_lds"] == (not force_global)
    rs = np.random.RandomState(lanes)
    for cls, model in ((0, m_ga), (1, m_nga)):
        xy = model.astype(np.float32)
        q = xy[rs.randint(0, len(xy), 300)] + rs.randn(300, 2).astype(np.float32) * \
            rs.choice([0.01, 0.3, 3.0, 30.0], size=(300, 1)).astype(np.float32)
        dis, idx = icp.nearest(cls, q)
        for k in range(len(q)):
            d, i = O.brute_nn1(xy, q[k, 0], q[k, 1])
            assert dis[k] == np.float32(d) and idx[k] == i, (cls, k, q[k])
    icp.close()


def test_nearest_ties_pick_lowest_index_and_empty_class():
    gx, gy = np.meshgrid(np.arange(12), np.arange(12))
    xy = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64)
    xy = np.concatenate([xy, xy[:30]])              # duplicates
    icp = api.Icp(np.zeros((0, 2)), xy)
    q = np.array([[0.5, 0.5], [3.5, 7.0], [5.0, 5.0], [-4, 3.5], [11.5, 11.5]], np.float32)
    dis, idx = icp.nearest(1, q)
    for k in range(len(q)):
        d, i = O.brute_nn1(xy.astype(np.float32), q[k, 0], q[k, 1])
        assert dis[k] == np.float32(d) and idx[k] == i
    dis, idx = icp.nearest(0, q)                     # empty class
    assert (idx == -1).all() and (dis == np.float32(1e38)).all()
    icp.close()


def test_config1_matches_golden_chain_and_oracle(world, golden_dir):
    """BASELINE config 1: one 1081-beam scan, 10k-point map, 20 iterations."""
    m_ga, m_nga, model = world
    G = np.load(os.path.join(golden_dir, "icp_chain_golden.npz"))
    batch = synth.make_batch(1, n_loop=256)
    icp = api.Icp(m_ga, m_nga)  # defaults: 20 iterations, 1e-6 (icp.cpp:27)
    R, t, res, trace = icp.fit_batch(batch, indist=5.0, trace=True)
    chain = G["chain"]
    steps = int(res["iters"][0])
    assert steps == len(chain)
    tr = trace[0, :steps]
    assert np.array_equal(tr[:, 7], chain[:, 7])                 # correspondences per step
    assert np.abs(tr[:, 4:6] - chain[:, 4:6]).max() < POS_TOL
    assert ang_diff(yaw(tr[:, :4]), yaw(chain[:, :4])).max() < ANG_TOL
    assert np.abs(tr[:, :6] - chain[:, :6]).max() < 1e-9        # in fact far tighter
    # host entry point, same answer
    t_ga, t_nga = batch.scan(0)
    R2, t2, r2 = icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
    assert np.array_equal(R2.reshape(4), R[0]) and np.array_equal(t2, t[0])
    assert (r2.iters, r2.n_corr) == (steps, int(res["n_corr"][0]))
    icp.close()


@pytest.mark.parametrize("lanes", [0, 1, 2, 8, 64, -1, -2])
def test_batch_matches_oracle(world, lanes):
    """32 scans of the config-2 loop, 30 iterations, fixed count (min_delta -1)."""
    m_ga, m_nga, model = world
    batch = synth.make_batch(32, n_loop=256)
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, lanes_per_point=lanes)
    R, t, res, _ = icp.fit_batch(batch, indist=5.0)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R,
                                                  batch.t, O.icp_params(30, -1.0, 5.0))
    assert np.array_equal(res["iters"], iters) and (iters == 30).all()
    assert np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < POS_TOL
    assert ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    assert np.abs(res["delta"] - delta).max() < 1e-9
    icp.close()


def test_early_exit_and_iteration_counts(world):
    m_ga, m_nga, model = world
    batch = synth.make_batch(8, n_loop=256)
    icp = api.Icp(m_ga, m_nga, max_iter=200, min_delta=1e-6)
    R, t, res, _ = icp.fit_batch(batch)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R,
                                                  batch.t, O.icp_params(200, 1e-6, 5.0))
    assert np.array_equal(res["iters"], iters) and (iters < 200).all()
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    # converged pose is the true pose to within the noise
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.03
    icp.close()


def test_setters_mirror_reference(world):
    m_ga, m_nga, model = world
    batch = synth.make_batch(1, n_loop=256)
    icp = api.Icp(m_ga, m_nga)
    icp.set_max_iterations(3)       # icp.h:51
    icp.set_min_delta(-1.0)         # icp.h:54
    t_ga, t_nga = batch.scan(0)
    R, t, res = icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
    assert res.iters == 3
    Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[0], batch.t[0], O.icp_params(3, -1.0, 5.0))
    assert np.abs(t - to).max() < 1e-9
    icp.close()


def test_edge_cases_follow_reference(world):
    m_ga, m_nga, model = world
    icp = api.Icp(m_ga, m_nga)
    R0, t0 = synth.pose_to_Rt(0.1, 0.2, 0.3)
    # icp.cpp:100-103: fewer than 5 template points -> error, R,t untouched
    with pytest.raises(api.SlamError) as e:
        icp.fit(np.zeros((2, 2)), np.zeros((2, 2)), R0, t0)
    assert e.value.code == api.E_TOO_FEW_SCENE
    # icpPointToPoint.cpp:128-131: no correspondence -> -1, loop stops, R,t unchanged
    far = np.full((10, 2), 500.0)
    R, t, res = icp.fit(far, np.zeros((0, 2)), R0, t0)
    assert (res.iters, res.n_corr, res.delta) == (1, 0, -1.0)
    assert np.array_equal(R, R0) and np.array_equal(t, t0)
    # batch: a short scan between two good ones is left untouched (ragged input)
    good = synth.make_batch(2, n_loop=256)
    pts = np.concatenate([good.scan(0)[0], good.scan(0)[1], np.zeros((3, 2)), good.scan(1)[0], good.scan(1)[1]])
    n0 = good.scan_off[1]
    off = np.array([0, n0, n0 + 3, n0 + 3 + (good.scan_off[2] - good.scan_off[1])], np.int32)
    nga = np.array([good.scan_nga[0], 1, good.scan_nga[1]], np.int32)
    Rb = np.stack([good.R[0], R0.reshape(4), good.R[1]])
    tb = np.stack([good.t[0], t0, good.t[1]])
    b3 = synth.ScanBatch(pts, off, nga, Rb, tb, np.zeros((3, 3)))
    R3, t3, res3, _ = icp.fit_batch(b3)
    assert res3["iters"][1] == 0 and np.array_equal(R3[1], R0.reshape(4)) and np.array_equal(t3[1], t0)
    Rg, tg, resg, _ = icp.fit_batch(good)
    # (few scans run in the spread form: a scan's sums are added over as many workgroups as the batch leaves it, so
    # the same scan in another batch agrees to rounding; with one workgroup per scan it agrees bit for bit)
    assert np.abs(R3[[0, 2]] - Rg).max() < 1e-12 and np.abs(t3[[0, 2]] - tg).max() < 1e-12
    assert np.array_equal(res3["n_corr"][[0, 2]], resg["n_corr"]) and np.array_equal(res3["iters"][[0, 2]], resg["iters"])
    icp.close()
    one = api.Icp(m_ga, m_nga, spread_scans=-1)
    R3, t3, _, _ = one.fit_batch(b3)
    Rg, tg, _, _ = one.fit_batch(good)
    assert np.array_equal(R3[[0, 2]], Rg) and np.array_equal(t3[[0, 2]], tg)
    one.close()


def test_class_constraint_and_small_class_skip():
    # icpPointToPoint.cpp:59,93: a class with <= 3 MODEL points is skipped entirely
    rs = np.random.RandomState(9)
    m_nga = rs.uniform(-5, 5, (200, 2))
    m_ga = rs.uniform(-5, 5, (3, 2))
    icp = api.Icp(m_ga, m_nga, max_iter=1, min_delta=-1)
    model = O.IcpModel(m_ga, m_nga)
    R0, t0 = synth.pose_to_Rt(0.01, 0.0, 0.0)
    t_ga, t_nga = m_ga + 0.01, m_nga[:50] + 0.01
    R, t, res = icp.fit(t_ga, t_nga, R0, t0)
    d, Ro, to, nc, corr = model.fit_step(t_ga, t_nga, R0, t0, O.icp_params())
    assert res.n_corr == nc == 50
    assert np.abs(t - to).max() < 1e-12 and np.abs(R - Ro).max() < 1e-12
    # scene GA points only ever match model GA points
    m_ga2 = rs.uniform(-5, 5, (50, 2))
    icp2 = api.Icp(m_ga2, m_nga + 100.0, max_iter=1, min_delta=-1)   # NGA model far away
    R, t, res = icp2.fit(np.zeros((0, 2)), m_ga2[:20] + 0.01, R0, t0)  # scene is all NGA
    assert res.n_corr == 0 and res.delta == -1.0
    icp.close(); icp2.close()


def test_all_nga_variant_and_global_memory_index(world):
    """nGA = 0 is legal (guarded at icpPointToPoint.cpp:59); and the HBM-resident
    index gives the same poses as the LDS-resident one."""
    m_ga, m_nga = synth.make_map(all_nga=True)
    assert len(m_ga) == 0
    batch = synth.make_batch(4, n_loop=256, all_nga=True)
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R,
                                                  batch.t, O.icp_params(20, 1e-6, 5.0))
    outs = []
    for fg in (0, 1):
        icp = api.Icp(m_ga, m_nga, force_global=fg)
        R, t, res, _ = icp.fit_batch(batch)
        assert np.array_equal(res["n_corr"], ncorr) and np.array_equal(res["iters"], iters)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
        outs.append((R, t))
        icp.close()
    # the two indexes use different lattices, so different queries take the cooperative path and the
    # sums run in another order: equal to rounding, not bitwise
    assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-9 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-9


def test_large_model_falls_back_to_hbm_index():
    """A model too large for 160 KB of LDS (2 x 19 999 points, the CCICP cap,
    icpTools.h:21) is served from HBM/L2 with identical results."""
    m_ga, m_nga = synth.make_map(39998)
    icp = api.Icp(m_ga, m_nga, max_iter=10, min_delta=-1)
    assert not icp.index_info()["in_lds"]
    batch = synth.make_batch(4, n_loop=256)
    R, t, res, _ = icp.fit_batch(batch)
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R,
                                                  batch.t, O.icp_params(10, -1, 5.0))
    assert np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    icp.close()


def test_config2_full_size_properties(world):
    """256 scans x 30 iterations (BASELINE config 2) -- size-independent checks:
    determinism, per-scan independence of the batch, convergence to the truth."""
    m_ga, m_nga, _ = world
    batch = synth.make_batch(256)
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0)
    R, t, res, _ = icp.fit_batch(batch)
    R2, t2, res2, _ = icp.fit_batch(batch)
    assert np.array_equal(R, R2) and np.array_equal(t, t2)            # bitwise reproducible
    sub = batch.shard(3, 8)                                           # scans 96..127 alone
    Rs, ts, _, _ = icp.fit_batch(sub)
    assert np.array_equal(Rs, R[96:128]) and np.array_equal(ts, t[96:128])
    assert (res["iters"] == 30).all() and (res["n_corr"] > 900).all()
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.03
    assert ang_diff(yaw(R), batch.true_poses[:, 2]).max() < 3e-3
    icp.close()


def test_config4_share_1024_scans_properties(world):
    """One GPU's share of BASELINE config 4: 1024 scans in one launch, four workgroups per CU in turn.  The same
    size-independent checks: determinism, independence of the batch it runs in (each quarter alone gives the same
    bits), convergence; and the early-exit form (min_delta 1e-6) against the oracle on a sample of the scans."""
    m_ga, m_nga, model = world
    batch = synth.make_batch(1024)
    # the library default pairs scans from two per CU on (two scans per workgroup, eight wavefronts each): within one form
    # a scan's result does not depend on the batch it runs in, bit for bit; between the forms the sums are grouped
    # differently (8 or 16 wavefronts), so the default is compared across batch sizes to rounding
    for pair in (-1, 2, 0):
        icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0, pair_scans=pair)
        R, t, res, _ = icp.fit_batch(batch)
        R2, t2, _, _ = icp.fit_batch(batch)
        assert np.array_equal(R, R2) and np.array_equal(t, t2)
        for q in range(4):
            Rq, tq, _, _ = icp.fit_batch(batch.shard(q, 4))
            if pair:
                assert np.array_equal(Rq, R[256 * q:256 * (q + 1)]) and np.array_equal(tq, t[256 * q:256 * (q + 1)]), (pair, q)
            else:
                assert np.abs(Rq - R[256 * q:256 * (q + 1)]).max() < 1e-8 and np.abs(tq - t[256 * q:256 * (q + 1)]).max() < 1e-8, q
        if pair:
            icp.close()
    assert (res["iters"] == 30).all() and (res["n_corr"] > 900).all()
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.03
    assert ang_diff(yaw(R), batch.true_poses[:, 2]).max() < 3e-3
    icp.close()
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=1e-6)
    R, t, res, _ = icp.fit_batch(batch)
    for s in range(0, 1024, 97):
        o, e, g = batch.scan_off[s], batch.scan_off[s + 1], batch.scan_nga[s]
        Ro, to, trace, steps = model.fit(batch.pts[o:o + g], batch.pts[o + g:e], batch.R[s].reshape(2, 2), batch.t[s],
                                         O.icp_params(30, 1e-6, 5.0))
        assert res["iters"][s] == steps, s
        assert np.abs(t[s] - to).max() < POS_TOL and abs(ang_diff(yaw(R[s:s + 1]), yaw(Ro.reshape(1, 4)))[0]) < ANG_TOL, s
    icp.close()


P2L_FORMS = {
    # what slam_icp_fit_batch_dev dispatches to (icp.hip launch_fit): (scans, parameters)
    "spread": (6, dict()),                                            # few scans: many workgroups per scan
    "fused": (40, dict(spread_scans=-1)),                             # ring search, then list sweeps, one scan per workgroup
    "pairs": (41, dict(spread_scans=-1, pair_scans=2)),               # the same with two scans per workgroup (odd batch)
    "ring-only": (20, dict(spread_scans=-1, lanes_per_point=2)),      # round 1's form: the ring search for every iteration
    "index-in-hbm": (20, dict(spread_scans=-1, force_global=1)),      # a model too large for LDS takes this
}


@pytest.mark.parametrize("form", sorted(P2L_FORMS))
def test_point_to_line_mode_matches_own_oracle(form):
    """north-star 3x3 normal-equation step (icpPointToPlane.cpp:37-107, not compiled upstream): normals and poses of
    every launch form against the build's own scalar oracle, delta of every step."""
    n_scans, kw = P2L_FORMS[form]
    m_ga, m_nga = synth.make_map(5000)
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10, max_iter=15, min_delta=-1.0, **kw)
    info = icp.index_info()
    assert info["two_forms"] == (form not in ("ring-only",)) and info["in_lds"] == (form != "index-in-hbm")
    n_gpu, n_cpu = icp.normals(), model.normals()
    # a normal and its negative are the same line: compare up to sign
    dots = np.abs((n_gpu * n_cpu).sum(1))
    assert dots.min() > 1 - 1e-9
    batch = synth.make_batch(n_scans, n_loop=256)
    R, t, res, trace = icp.fit_batch(batch, trace=True)
    for s in range(batch.n_scans):
        t_ga, t_nga = batch.scan(s)
        Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[s], batch.t[s],
                                      O.icp_params(15, -1.0, 5.0, O.NN_KDTREE, O.MODE_P2L))
        assert steps == res["iters"][s] == 15
        assert res["n_corr"][s] == len(t_ga) + len(t_nga)          # every template point (icpPointToPlane.cpp:55-77)
        assert np.abs(t[s] - to).max() < POS_TOL
        assert ang_diff(yaw(R[s]), yaw(Ro)).max() < ANG_TOL
        if np.abs(trace[s, :, 6] - tr[:, 6]).max() >= 1e-7:
            # an exact float distance tie somewhere in the scan (scan 14 of this batch has one in its first step: two model
            # points at 0.009734867 m^2 of one query): the kd-tree takes the one it visits last (kdtree.cpp:612-618), the
            # GPU and the brute-force arbiter (kdtree.cpp:360-375) the lowest index -- the arbiter decides
            _, _, tr, _ = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], O.icp_params(15, -1.0, 5.0, O.NN_BRUTE, O.MODE_P2L))
        assert np.abs(trace[s, :, 6] - tr[:, 6]).max() < 1e-7, s
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.03
    icp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pairs", [False, True])
def test_point_to_line_hand_over_edge_iteration_counts(pairs):
    """Point-to-line scans may hand over to the list form after TWO first iterations (round 5; the guard decides): max_iter below,
    at and just above that, an explicit later hand-over, and min_delta reached in either form -- steps, poses and the delta of
    every executed step against the oracle, one scan and two scans per workgroup."""
    m_ga, m_nga = synth.make_map(5000)
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    batch = synth.make_batch(7, n_loop=256)
    form = dict(spread_scans=-1, pair_scans=2 if pairs else -1)
    for max_iter, min_delta, first in ((1, -1.0, 0), (2, -1.0, 0), (3, -1.0, 0), (4, -1.0, 0), (12, -1.0, 6), (40, 1e-3, 0), (40, 1e-7, 0)):
        icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10, max_iter=max_iter, min_delta=min_delta, first_iterations=first, **form)
        info = icp.index_info()
        assert info["two_forms"] and info["first_iterations"] == (first or 2)
        R, t, res, trace = icp.fit_batch(batch, trace=True)
        for s in range(batch.n_scans):
            t_ga, t_nga = batch.scan(s)
            Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], O.icp_params(max_iter, min_delta, 5.0, O.NN_KDTREE, O.MODE_P2L))
            if res["iters"][s] != steps or np.abs(trace[s, :steps, 6] - tr[:steps, 6]).max() >= 1e-7:   # an exact distance tie: the arbiter
                Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], O.icp_params(max_iter, min_delta, 5.0, O.NN_BRUTE, O.MODE_P2L))
            assert res["iters"][s] == steps, (max_iter, min_delta, s)
            assert np.abs(trace[s, :steps, 6] - tr[:steps, 6]).max() < 1e-7, (max_iter, min_delta, s)
            assert np.abs(t[s] - to).max() < POS_TOL and ang_diff(yaw(R[s]), yaw(Ro)).max() < ANG_TOL
        icp.close()


def test_point_to_line_host_fit_and_early_exit():
    """Icp::fit's shape (icp.h:65) in point-to-line mode: one scan through slam_icp_fit (the spread form), stopping on
    min_delta after the same number of steps as the oracle; a template of fewer than 5 points leaves R, t untouched."""
    m_ga, m_nga = synth.make_map(5000)
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10, max_iter=40, min_delta=1e-6)
    batch = synth.make_batch(3, n_loop=256)
    for s in range(3):
        t_ga, t_nga = batch.scan(s)
        R, t, res = icp.fit(t_ga, t_nga, batch.R[s], batch.t[s], 5.0)
        Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], O.icp_params(40, 1e-6, 5.0, O.NN_KDTREE, O.MODE_P2L))
        assert res.iters == steps and steps < 40
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    t_ga, t_nga = batch.scan(0)
    with pytest.raises(api.SlamError):
        icp.fit(t_ga[:2], t_nga[:2], batch.R[0], batch.t[0], 5.0)
    icp.close()


def test_point_to_line_model_from_device_arrays_and_through_the_mapper():
    """slam_icp_create_dev in point-to-line mode (the model's f64 arrays already in HBM): the same normals and the same fits,
    bit for bit, as the handle made from host arrays; and the streaming mapper with a fixed point-to-line target registers its
    chunks as the batch call does -- and with a SLIDING one, whose deferred rebuilds make their normals on the device."""
    m_ga, m_nga = synth.make_map(6000)
    batch = synth.make_batch(24, n_loop=256)
    kw = dict(mode=api.ICP_P2L, normals_k=10, max_iter=12, min_delta=-1.0)
    host = api.Icp(m_ga, m_nga, **kw)
    d_ga, d_nga = api.DeviceArray.from_host(m_ga, np.float64), api.DeviceArray.from_host(m_nga, np.float64)
    dev = api.Icp.from_device(d_ga, len(m_ga), d_nga, len(m_nga), **kw)
    assert np.array_equal(host.normals(), dev.normals())
    Rh, th, rh, _ = host.fit_batch(batch)
    Rd, td, rd, _ = dev.fit_batch(batch)
    assert np.array_equal(Rh, Rd) and np.array_equal(th, td) and np.array_equal(rh, rd)
    host.close()
    dev.close()
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=500, grid_size_y=500, resolution=0.1,
                    max_scans=batch.n_scans, max_points=batch.n_points, icp=kw)
    Rm, tm = mp.wait(mp.push(batch))
    mp.finish()
    mp.close()
    assert np.abs(Rm - Rh).max() < 1e-9 and np.abs(tm - th).max() < 1e-9       # (the mapper's batch may take another launch form)
    assert np.abs(tm - batch.true_poses[:, :2]).max() < 0.03
    # a SLIDING point-to-line target: the deferred rebuild merges the classes and makes the normals on the device (round 4)
    mp2 = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=500, grid_size_y=500, resolution=0.1,
                     max_scans=batch.n_scans, max_points=batch.n_points + 1000, icp=kw, window_chunks=2, rebuild_every=1, thin_res=0.1,
                     keep_prior=1, target_points=4000, strict_window=1)
    for k in range(4):
        b = synth.make_batch(24, n_loop=256, first=24 * k)
        Rs, ts = mp2.wait(mp2.push(b))
        assert np.abs(ts - b.true_poses[:, :2]).max() < 0.03, k
    mp2.finish()
    assert mp2.stats()["rebuilds"] >= 3
    info = mp2.target_index_info()
    assert info["two_forms"] and info["built_on_device"]
    mp2.close()


def test_point_to_line_edge_cases_follow_the_oracle():
    """icpPointToPlane.cpp:37-107 at its edges: a template below 5 points leaves R, t alone (icp.cpp:100-103); a model and
    templates of one class only; a model that is ONE straight wall -- every normal the same, the translation along the wall
    unobservable, A^T A singular or nearly so: whatever the reference's Gauss-Jordan makes of it (a refused solve leaves the pose
    and returns 0, icpPointToPlane.cpp:85), the GPU makes the same of it, step count included."""
    rs = np.random.RandomState(11)
    # (a) ordinary small model, all points of class NGA; scans of 5 and of 4 points
    m_nga = np.stack([np.linspace(-5, 5, 400), 2.0 + 0.01 * rs.randn(400)], 1)
    m_nga = np.concatenate([m_nga, np.stack([4.0 + 0.01 * rs.randn(300), np.linspace(-4, 2, 300)], 1)])
    m_ga = np.zeros((0, 2))
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    prm = O.icp_params(12, 1e-6, 5.0, O.NN_BRUTE, O.MODE_P2L)
    icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10, max_iter=12, min_delta=1e-6)
    th = 0.03
    R0, t0 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]), np.array([0.05, -0.04])
    scan = np.concatenate([m_nga[::90][:4] + 0.002 * rs.randn(4, 2), m_nga[420::60][:3] + 0.002 * rs.randn(3, 2)])
    for t_ga, t_nga in ((scan[:0], scan), (scan[:3], scan[3:]), (scan[:5], scan[:0])):
        R, t, res = icp.fit(t_ga, t_nga, R0, t0, 5.0)
        Ro, to, tr, steps = model.fit(t_ga, t_nga, R0, t0, prm)
        assert res.iters == steps and res.n_corr == len(t_ga) + len(t_nga)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    with pytest.raises(api.SlamError):
        icp.fit(scan[:2], scan[2:4], R0, t0, 5.0)
    icp.close()
    # (b) one straight wall: every normal is (0, 1) up to sign, the x translation is unobservable
    wall = np.stack([np.linspace(-3, 3, 200), np.full(200, 1.0)], 1)
    model = O.IcpModel(wall[:0], wall, normals_k=10)
    icp = api.Icp(wall[:0], wall, mode=api.ICP_P2L, normals_k=10, max_iter=12, min_delta=1e-6)
    assert np.abs(np.abs(icp.normals()[:, 1]) - 1.0).max() < 1e-12
    sc = wall[20:180:8] + [0.0, 0.07]
    R, t, res = icp.fit(sc[:0], sc, np.eye(2), np.zeros(2), 5.0)
    Ro, to, tr, steps = model.fit(sc[:0], sc, np.eye(2), np.zeros(2), prm)
    assert res.iters == steps
    assert np.allclose(t, to, atol=1e-9) and np.allclose(R.reshape(2, 2), Ro, atol=1e-9)
    icp.close()


def test_edge_weight_matches_oracle(world):
    """getEdgeWeight (icpPointToPoint.cpp:233-316) over the correspondences of the
    last executed fitStep, reference bug (dy = ax - bx) included."""
    m_ga, m_nga, model = world
    batch = synth.make_batch(1, n_loop=256)
    t_ga, t_nga = batch.scan(0)
    icp = api.Icp(m_ga, m_nga, max_iter=7, min_delta=-1.0)
    R, t, res = icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
    eW = icp.edge_weight()
    p = O.icp_params(7, -1.0, 5.0)
    Ro, to, tr, steps = model.fit(t_ga, t_nga, batch.R[0], batch.t[0], p)
    Rp, tp = tr[steps - 2, :4].reshape(2, 2), tr[steps - 2, 4:6]      # pose the last step started from
    d, _, _, nc, corr = model.fit_step(t_ga, t_nga, Rp, tp, p)
    q = np.concatenate([O.transform_points(t_ga, Rp, tp), O.transform_points(t_nga, Rp, tp)]).astype(np.float64)
    mga32, mnga32 = m_ga.astype(np.float32).astype(np.float64), m_nga.astype(np.float32).astype(np.float64)
    is_ga = np.arange(len(corr)) < len(t_ga)
    sel = corr >= 0
    pm = np.where(is_ga[sel, None], mga32[np.clip(corr[sel], 0, len(mga32) - 1)],
                  mnga32[np.clip(corr[sel], 0, len(mnga32) - 1)])
    ref = O.edge_weight(pm, q[sel])
    assert nc == res.n_corr and ref[0, 0] == ref[1, 1] and ref[0, 1] == 0
    assert np.allclose(eW, ref, rtol=1e-7, atol=1e-7 * np.abs(ref).max())
    icp.close()


@pytest.mark.parametrize("lanes,spread", [(0, 0), (0, -1), (2, 0), (-2, 0)])
def test_ragged_scan_sizes(world, lanes, spread):
    """Scans from 5 to ~2600 points in one batch: the pass structure (full passes, short tails, the
    cooperative queue of the sweep mode, scans smaller than one cooperative round) against the oracle.
    (0, 0): twelve scans are few enough for the spread form (21 workgroups per scan, 64 lanes per query);
    (0, -1): the same batch with one workgroup per scan, ring search then list sweeps."""
    m_ga, m_nga, model = world
    pts, off, nga, Rs, ts = [], [0], [], [], []
    for k, beams in enumerate([6, 41, 64, 66, 700, 1026, 1027, 1081, 1090, 1100, 2200, 2600]):
        ga, ng, pose = synth.make_scan(3 * k, 256, n_beams=beams)
        pts += [ga, ng]
        off.append(off[-1] + len(ga) + len(ng))
        nga.append(len(ga))
        R, t = synth.pose_to_Rt(*synth.init_pose(3 * k, pose))
        Rs.append(R.reshape(4))
        ts.append(t)
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(pts)), np.array(off, np.int32), np.array(nga, np.int32),
                            np.array(Rs), np.array(ts), np.zeros((len(nga), 3)))
    sizes = np.diff(batch.scan_off)
    assert sizes.min() >= 5 and sizes.max() > 2 * 1024 + 64
    icp = api.Icp(m_ga, m_nga, max_iter=12, min_delta=-1.0, lanes_per_point=lanes, spread_scans=spread)
    R, t, res, _ = icp.fit_batch(batch, indist=5.0)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(12, -1.0, 5.0))
    assert np.array_equal(res["iters"], iters) and np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    R2, t2, _, _ = icp.fit_batch(batch, indist=5.0)
    assert np.array_equal(R, R2) and np.array_equal(t, t2)     # bitwise reproducible
    icp.close()


@pytest.mark.parametrize("frac", [0.01, 0.1, 0.5])
def test_outlier_scans_two_form_schedule(world, frac):
    """Scans with points far from the map (some inside the 5.0 gate, some beyond it): few of them and the scan is
    handed to the list sweeps with a busy cooperative queue, many and it stays in the ring search (the
    hand-over guard); either way the result is the oracle's."""
    m_ga, m_nga, model = world
    batch = synth.make_batch(6, n_loop=256)
    rs = np.random.RandomState(int(frac * 1000))
    pts = batch.pts.copy()
    for s in range(batch.n_scans):
        o, e = batch.scan_off[s], batch.scan_off[s + 1]
        k = rs.choice(np.arange(o, e), int(frac * (e - o)), replace=False)
        pts[k] += rs.uniform(-4.0, 4.0, (len(k), 2))           # up to 4 m off: neighbours at 0.2 .. 4 m
    noisy = synth.ScanBatch(pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, batch.true_poses)
    Ro, to, iters, ncorr, delta = model.fit_batch(noisy.pts, noisy.scan_off, noisy.scan_nga, noisy.R, noisy.t,
                                                  O.icp_params(25, 1e-7, 5.0))
    for lanes, spread in ((0, -1), (0, 0), (2, 0), (-2, 0)):   # (0, -1): the two-form schedule; (0, 0): the spread form
        icp = api.Icp(m_ga, m_nga, max_iter=25, min_delta=1e-7, lanes_per_point=lanes, spread_scans=spread)
        R, t, res, _ = icp.fit_batch(noisy, indist=5.0)
        assert np.array_equal(res["iters"], iters) and np.array_equal(res["n_corr"], ncorr), (lanes, spread)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL, (lanes, spread)
        icp.close()


@pytest.mark.parametrize("cell", [0.33, 0.5, 1.3])
def test_cell_pitch_is_only_a_cost_parameter(world, cell):
    """The lattice pitch changes the cost of the exact search, never its result: nearest neighbours bit-exact,
    batch fit as the oracle (tools/cell_sweep.sh measures the cost side)."""
    m_ga, m_nga, model = world
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=1e-6, cell_size=cell)
    assert abs(icp.index_info()["cell"] - cell) < 1e-6
    rs = np.random.RandomState(3)
    xy = m_nga.astype(np.float32)
    q = xy[rs.randint(0, len(xy), 200)] + rs.randn(200, 2).astype(np.float32) * \
        rs.choice([0.01, 0.3, 3.0], size=(200, 1)).astype(np.float32)
    dis, idx = icp.nearest(1, q)
    for k in range(len(q)):
        d, i = O.brute_nn1(xy, q[k, 0], q[k, 1])
        assert dis[k] == np.float32(d) and idx[k] == i, (k, q[k])
    batch = synth.make_batch(12, n_loop=256)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(30, 1e-6, 5.0))
    R, t, res, _ = icp.fit_batch(batch, indist=5.0)
    assert np.array_equal(res["iters"], iters) and np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    icp.close()


def test_one_launch_and_two_launch_schedules_agree_bitwise(world):
    """The default runs both search forms in one launch (the workgroup swaps its LDS contents); split_launch = 1
    runs them as two launches with the hand-over state in HBM.  Same arithmetic, same order: identical bits."""
    m_ga, m_nga, model = world
    batch = synth.make_batch(24, n_loop=256)
    out = []
    for split in (0, 1):
        icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=1e-6, split_launch=split)
        R, t, res, _ = icp.fit_batch(batch, indist=5.0)
        out.append((R.copy(), t.copy(), res.copy()))
        icp.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


def test_two_form_schedule_edge_iteration_counts(world):
    """max_iter below, at and just above the hand-over iteration; min_delta reached in either form."""
    m_ga, m_nga, model = world
    batch = synth.make_batch(5, n_loop=256)
    for max_iter, min_delta in ((1, -1.0), (9, -1.0), (10, -1.0), (11, -1.0), (40, 1e-2), (40, 1e-4), (40, 1e-9)):
        Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                      O.icp_params(max_iter, min_delta, 5.0))
        icp = api.Icp(m_ga, m_nga, max_iter=max_iter, min_delta=min_delta, spread_scans=-1)   # one workgroup per scan
        assert icp.index_info()["two_forms"] and icp.index_info()["first_iterations"] == 10
        R, t, res, _ = icp.fit_batch(batch, indist=5.0)
        assert np.array_equal(res["iters"], iters), (max_iter, min_delta)
        assert np.array_equal(res["n_corr"], ncorr)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
        assert np.abs(res["delta"] - delta).max() < 1e-9
        icp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["fused", "pairs", "ring", "lists", "per_pass"])
def test_exact_ties_through_every_batch_form(form):
    """A gridded model with duplicated points, queries that sit exactly between four model points (identity pose: the ties are
    exact in float) or on a model point that exists twice: the fast scans notice a tie as second best == best (and a seeded search
    meets its own seed again one ulp below its starting value, which is no tie), the exact pass settles it by the lowest ORIGINAL
    index -- the brute-force arbiter's rule (kdtree.cpp:360-375).  Every launch form of the batch kernels, step by step."""
    from test_gpu_icp_spread import check_against_oracle
    kw = {"fused": dict(spread_scans=-1), "pairs": dict(spread_scans=-1, pair_scans=2), "ring": dict(spread_scans=-1, lanes_per_point=2),
          "lists": dict(spread_scans=-1, lanes_per_point=-2), "per_pass": dict(spread_scans=-1, lanes_per_point=-1)}[form]
    rs = np.random.RandomState(23)
    gx, gy = np.meshgrid(np.arange(60) * 0.5, np.arange(40) * 0.5)
    grid = np.stack([gx.ravel(), gy.ravel()], 1)
    m_nga = np.concatenate([grid, grid[:600]])                 # the first 600 grid points exist twice
    m_ga = np.concatenate([grid[1000:1400] + [0.25, 0.0], grid[1000:1100] + [0.25, 0.0]])
    scans, nga, Rs, ts = [], [], [], []
    for k in range(24):
        kind = k % 3
        pick = grid[rs.choice(len(grid), 400)]
        if kind == 0:
            pts, pose = pick + 0.25, (0.0, 0.0, 0.0)           # equidistant from four model points, identity pose
        elif kind == 1:
            pts, pose = pick.copy(), (0.0, 0.0, 0.0)           # on model points (some of them duplicated), identity pose
        else:
            pts, pose = pick + rs.randn(400, 2) * 0.03, (0.05, -0.04, 0.004)
        n_ga = 60 if k % 2 else 0                                 # some scans carry class-GA points (their own, smaller model)
        R0, t0 = synth.pose_to_Rt(*pose)
        scans.append(pts); nga.append(n_ga); Rs.append(R0.reshape(4)); ts.append(t0)
    off = np.cumsum([0] + [len(x) for x in scans]).astype(np.int32)
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(scans)), off, np.array(nga, np.int32), np.array(Rs), np.array(ts),
                            np.zeros((len(scans), 3)))
    icp, R, t, res, tr = check_against_oracle(m_ga, m_nga, batch, 12, 1e-9, nn=O.NN_BRUTE, **kw)
    assert (res["n_corr"] > 300).all()
    icp.close()


@pytest.mark.parametrize("map_points", [3000, 5000])
def test_list_lattice_choice_changes_time_not_results(map_points):
    """slam_icp_params::list_min_halo (round 5): WHICH halo-list lattice is built -- the finest that fits LDS (< 0, the rule of
    rounds 1-4), the first whose halo reaches 0.125 m (the default), a coarse one -- is a schedule decision: every form of the search
    is exact, so iterations and correspondence counts are equal and poses agree to the rounding of differently grouped sums, and all
    of them equal the oracle.  Models sparser than config 2's are where the choice differs (a 10 k-point room gets the same lattice
    from all three)."""
    m_ga, m_nga = synth.make_map(map_points)
    model = O.IcpModel(m_ga, m_nga)
    batch = synth.make_batch(48)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, O.icp_params(30, 1e-6, 5.0))
    seen, ref = set(), None
    for halo in (-1.0, 0.0, 0.3):
        for pair in (-1, 2):
            icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=1e-6, spread_scans=-1, pair_scans=pair, list_min_halo=halo)
            info = icp.index_info()
            assert info["two_forms"] and info["list_halo"] >= (halo if halo > 0 else (0.125 if halo == 0 else 0.0)) - 1e-6
            seen.add(round(info["list_pitch"], 4))
            R, t, res, _ = icp.fit_batch(batch)
            assert np.array_equal(res["iters"], iters) and np.array_equal(res["n_corr"], ncorr), (halo, pair)
            assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL, (halo, pair)
            if ref is None:
                ref = (R, t)
            assert np.abs(R - ref[0]).max() < 1e-9 and np.abs(t - ref[1]).max() < 1e-9, (halo, pair)
            icp.close()
    assert len(seen) >= 2, "the three settings built the same lattice: nothing was tested (%s)" % seen


@pytest.mark.gpu
@pytest.mark.parametrize("n", [600, 12000, 39998])
def test_point_to_line_normals_of_scattered_models(n):
    """Normals by k-NN where the tenth neighbour is many cells away: scattered points (no walls), a dense blob, stragglers far out in
    the fringe -- the search's squares grow past their first few radii and read side spans beside what they have seen (round 6;
    a room's walls keep it within two cells) -- against the oracle's normals up to sign, and the handle made from device arrays."""
    rs = np.random.RandomState(n)
    pts = np.concatenate([rs.randn(n // 2, 2) * [30.0, 20.0], rs.rand(n - n // 2 - 40, 2) * [80.0, 60.0] - [40.0, 30.0],
                          rs.randn(30, 2) * 0.02 + [5.0, 5.0], rs.rand(10, 2) * 600.0 - 300.0])
    pts = np.ascontiguousarray(pts[rs.permutation(len(pts))])
    m_ga, m_nga = pts[: len(pts) // 3], pts[len(pts) // 3:]
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10)
    n_gpu, n_cpu = icp.normals(), model.normals()
    dots = np.abs((n_gpu * n_cpu).sum(1))
    # (ten points that are all but collinear or all but isotropic leave the direction to the last bits: allow a handful)
    assert np.sort(dots)[5] > 1 - 1e-9 and dots.min() > 0.99, (np.sort(dots)[:8])
    icp.close()
