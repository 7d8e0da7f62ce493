"""The spread form of the ICP (slam_amd/csrc/icp_single.hip): few scans -- one, in the reference's own usage
(scan_registration.cpp:139-159 -> IcpPointToPoint::fit, icp.cpp:80-122) -- each dealt over many workgroups of one
persistent launch that exchange their nine sums once per iteration.  Against the oracle and against the
one-workgroup-per-scan kernels; same tolerances as tests/test_gpu_icp.py."""
import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth

pytestmark = pytest.mark.gpu

POS_TOL, ANG_TOL = 1e-4, 1e-5


def yaw(R):
    R = np.asarray(R).reshape(-1, 4)
    return np.arctan2(R[:, 2], R[:, 0])


def ang_diff(a, b):
    d = a - b
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


def check_against_oracle(m_ga, m_nga, batch, max_iter, min_delta, nn=O.NN_KDTREE, **kw):
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(max_iter, min_delta, 5.0, nn))
    icp = api.Icp(m_ga, m_nga, max_iter=max_iter, min_delta=min_delta, **kw)
    R, t, res, tr = icp.fit_batch(batch, indist=5.0, trace=True)
    assert np.array_equal(res["iters"], iters), (res["iters"], iters)
    assert np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    assert np.abs(res["delta"] - delta).max() < 1e-9
    R2, t2, res2, _ = icp.fit_batch(batch, indist=5.0)
    assert np.array_equal(R, R2) and np.array_equal(t, t2)          # bitwise reproducible
    return icp, R, t, res, tr


@pytest.mark.parametrize("n_scans", [1, 3, 16])
@pytest.mark.parametrize("force_global", [0, 1])
def test_spread_matches_oracle_and_batch_kernels(n_scans, force_global):
    """Model in LDS (every workgroup copies it) and in HBM/L2; 1, 3 and 16 scans (256, 85, 16 workgroups each)."""
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(n_scans, n_loop=256)
    icp, R, t, res, tr = check_against_oracle(m_ga, m_nga, batch, 30, 1e-6, force_global=force_global)
    icp.close()
    # the same scans with one workgroup per scan: the same correspondences, the sums in another order
    ref = api.Icp(m_ga, m_nga, max_iter=30, min_delta=1e-6, force_global=force_global, spread_scans=-1)
    Rb, tb, resb, _ = ref.fit_batch(batch, indist=5.0)
    assert np.array_equal(resb["iters"], res["iters"]) and np.array_equal(resb["n_corr"], res["n_corr"])
    assert np.abs(tb - t).max() < 1e-9 and np.abs(Rb - R).max() < 1e-9
    ref.close()


def test_spread_trace_follows_the_oracle_step_by_step():
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(1, n_loop=256)
    model = O.IcpModel(m_ga, m_nga)
    t_ga, t_nga = batch.scan(0)
    Ro, to, otr, steps = model.fit(t_ga, t_nga, batch.R[0].reshape(2, 2), batch.t[0], O.icp_params(20, 1e-6, 5.0))
    icp = api.Icp(m_ga, m_nga)
    R, t, res, tr = icp.fit_batch(batch, trace=True)
    assert res["iters"][0] == steps
    for k in range(steps):
        assert np.abs(tr[0, k, :6] - otr[k, :6]).max() < 1e-9, k
        assert tr[0, k, 7] == otr[k, 7]
    icp.close()


def test_spread_large_model_and_large_scans():
    """2 x 19 999 model points (index in HBM, 32-bit starts); scans of 1081, 4096, 4097 (16 lanes per query from there
    on) and 19 999 + 3 000 points (the CCICP cap per class, icpTools.h:21), class-constrained."""
    m_ga, m_nga = synth.make_map(39998)
    rs = np.random.RandomState(5)
    pts, off, nga, Rs, ts = [], [0], [], [], []
    for k, (n_g, n_n) in enumerate([(150, 931), (1000, 3096), (1000, 3097), (3000, 19999)]):
        th = 0.03 * (k + 1)
        Rt, tt = synth.pose_to_Rt(0.25 - 0.1 * k, -0.2 + 0.1 * k, th)
        Rt = Rt.reshape(2, 2)
        # scene = model points seen from the pose (x = R^T (m - t)) plus noise
        ga = (m_ga[rs.choice(len(m_ga), n_g, replace=n_g > len(m_ga))] - tt) @ Rt + rs.randn(n_g, 2) * 0.01
        ng = (m_nga[rs.choice(len(m_nga), n_n, replace=n_n > len(m_nga))] - tt) @ Rt + rs.randn(n_n, 2) * 0.01
        pts += [ga, ng]
        off.append(off[-1] + n_g + n_n)
        nga.append(n_g)
        R0, t0 = synth.pose_to_Rt(0.25 - 0.1 * k + 0.2, -0.2 + 0.1 * k - 0.15, th + 0.03)
        Rs.append(R0.reshape(4))
        ts.append(t0)
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(pts)), np.array(off, np.int32), np.array(nga, np.int32),
                            np.array(Rs), np.array(ts), np.zeros((4, 3)))
    icp, R, t, res, _ = check_against_oracle(m_ga, m_nga, batch, 25, 1e-6)
    assert not icp.index_info()["in_lds"]
    icp.close()


def test_spread_edge_cases():
    """Ragged batch: a scan below 5 points is left untouched (icp.cpp:100-103), a scan with no correspondence stops at
    delta -1 with its pose unchanged (icpPointToPoint.cpp:128-131), a class with <= 3 model points is skipped (:59,93);
    exact distance ties (gridded model with duplicates) take the exact pass."""
    rs = np.random.RandomState(11)
    gx, gy = np.meshgrid(np.arange(60) * 0.5, np.arange(40) * 0.5)
    grid = np.stack([gx.ravel(), gy.ravel()], 1)
    m_nga = np.concatenate([grid, grid[:200]])
    m_ga = grid[:3] + 0.1                                  # three points: the class is skipped
    R0, t0 = synth.pose_to_Rt(0.1, -0.05, 0.01)
    scans = [grid[rs.choice(len(grid), 500)] + 0.25,         # every query is equidistant from four model points
             np.zeros((3, 2)),                              # too short
             np.full((10, 2), 900.0),                       # nothing within the gate
             grid[rs.choice(len(grid), 64)] + rs.randn(64, 2) * 0.02]
    nga = [0, 1, 0, 10]                                     # the last scan's first 10 points are class GA: ignored
    off = np.cumsum([0] + [len(x) for x in scans]).astype(np.int32)
    Rs, ts = np.tile(R0.reshape(4), (4, 1)), np.tile(t0, (4, 1))
    Rs[0], ts[0] = [1, 0, 0, 1], [0, 0]                     # identity: the ties of scan 0 are exact in float
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(scans)), off, np.array(nga, np.int32),
                            Rs, ts, np.zeros((4, 3)))
    # exact ties: the reference leaves them to the kd-tree's visit order (kdtree.cpp:612-618); the brute-force
    # arbiter (:360-375) and the GPU take the lowest original index
    icp, R, t, res, _ = check_against_oracle(m_ga, m_nga, batch, 15, 1e-6, nn=O.NN_BRUTE)
    assert res["iters"][1] == 0 and np.array_equal(R[1], R0.reshape(4)) and np.array_equal(t[1], t0)
    assert (res["iters"][2], res["n_corr"][2], res["delta"][2]) == (1, 0, -1.0)
    assert np.array_equal(R[2], R0.reshape(4)) and np.array_equal(t[2], t0)
    assert res["n_corr"][3] == 54
    icp.close()


def test_host_fit_goes_through_the_spread_form_and_keeps_edge_weights():
    """slam_icp_fit (Icp::fit, host arrays) + getEdgeWeight of its last step, model too large for LDS."""
    m_ga, m_nga = synth.make_map(39998)
    batch = synth.make_batch(1, n_loop=256)
    t_ga, t_nga = batch.scan(0)
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, otr, steps = model.fit(t_ga, t_nga, batch.R[0].reshape(2, 2), batch.t[0], O.icp_params(20, 1e-6, 5.0))
    icp = api.Icp(m_ga, m_nga)
    R, t, res = icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
    assert res.iters == steps
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro))[0] < ANG_TOL
    eW = icp.edge_weight()
    assert np.isfinite(eW).all() and eW[0, 0] > 0
    icp.close()


# ---- a fit always returns a pose (Icp::fit, icp.cpp:80-114): the spread form needs all workgroups of a scan resident
# together, and when that does not happen the scan is redone by the one-workgroup form inside the same call

@pytest.mark.parametrize("n_scans,big", [(1, False), (5, False), (2, True)])
def test_spread_hand_over_redoes_the_scans(n_scans, big):
    """spread_wait_us < 0: every scan is handed over at its first exchange.  What comes back is the one-workgroup form's
    result, bit for bit, and the oracle's; model in LDS and (big) in HBM."""
    m_ga, m_nga = synth.make_map(39998 if big else 10000)
    batch = synth.make_batch(n_scans, n_loop=256)
    icp, R, t, res, tr = check_against_oracle(m_ga, m_nga, batch, 20, 1e-6, spread_wait_us=-1)
    icp.close()
    ref = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6, spread_scans=-1)
    Rb, tb, resb, trb = ref.fit_batch(batch, indist=5.0, trace=True)
    assert np.array_equal(R, Rb) and np.array_equal(t, tb) and np.array_equal(res, resb)
    ref.close()
    # the host API of one scan (slam_icp_fit): the same
    t_ga, t_nga = batch.scan(0)
    a = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6, spread_wait_us=-1)
    Ra, ta, ra = a.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
    assert np.array_equal(Ra.reshape(4), R[0]) and np.array_equal(ta, t[0]) and ra.iters == res["iters"][0]
    assert np.isfinite(a.edge_weight()).all()
    # ... and a handle whose spread launch was handed over does its next fits without the spread form (16 of them, then it tries
    # again: slam_icp::spread_backoff): forty fits in a row, through both forms as they alternate, all the same bits
    for rep in range(40):
        Rr, tr_, rr = a.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
        assert np.array_equal(Rr, Ra) and np.array_equal(tr_, ta) and rr.iters == ra.iters and rr.n_corr == ra.n_corr, rep
    assert np.isfinite(a.edge_weight()).all()
    a.close()


def test_two_handles_fit_from_two_threads():
    """The reference runs two CCICP objects in one process (scan_registration.cpp:57, graphSlamTools.cpp:14): two handles,
    two host threads, one fit after the other each.  Every fit equals the oracle and nothing stalls."""
    import threading
    import time
    m_ga, m_nga = synth.make_map(10000)
    batch = synth.make_batch(8, n_loop=256)
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, _ = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, O.icp_params(20, 1e-6, 5.0))
    handles = [api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6) for _ in range(2)]
    out = [[], []]

    def work(k):
        for rep in range(6):
            for s in range(k, batch.n_scans, 2):
                t_ga, t_nga = batch.scan(s)
                out[k].append((s,) + tuple(handles[k].fit(t_ga, t_nga, batch.R[s], batch.t[s], 5.0)))
    for h in handles:                                      # warm-up: buffers, code objects
        h.fit(*batch.scan(0), batch.R[0], batch.t[0], 5.0)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    assert len(out[0]) + len(out[1]) == 6 * batch.n_scans
    for k in range(2):
        for s, R, t, r in out[k]:
            assert r.iters == iters[s] and r.n_corr == ncorr[s]
            assert np.abs(t - to[s]).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro[s]))[0] < ANG_TOL
    assert dt < 1.0, "48 fits took %.3f s: a spread launch waited out its time limit" % dt
    [h.close() for h in handles]


def test_fit_beside_a_running_grid_update():
    """local_mapper runs beside scan_registration on the same machine: single fits while a stream of config 4's share of
    raycasts (1024 scans into 4000 x 4000, persistent workgroups on every CU) keeps the chip busy.  Every fit equals the
    oracle, none waits out a time limit."""
    import time
    m_ga, m_nga = synth.make_map(10000)
    big = synth.make_batch(1024, n_loop=1024)
    R = np.stack([synth.pose_to_Rt(*p)[0].reshape(4) for p in big.true_poses])
    t = np.stack([synth.pose_to_Rt(*p)[1] for p in big.true_poses])
    d = [api.DeviceArray.from_host(a, dt) for a, dt in ((big.pts, np.float64), (big.scan_off, np.int32), (R, np.float64), (t, np.float64))]
    g = api.Grid(4000, 4000, 0.05, rolling=0, min_cluster_points=20)
    st = api.Stream()
    batch = synth.make_batch(6, n_loop=256)
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, _ = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, O.icp_params(20, 1e-6, 5.0))
    icp = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6)
    icp.fit(*batch.scan(0), batch.R[0], batch.t[0], 5.0)
    g.raycast_scans_dev(d[0], d[1], big.n_scans, big.n_points, d[2], d[3], st)
    st.synchronize()
    t0 = time.perf_counter()
    for rep in range(4):
        for k in range(8):
            g.raycast_scans_dev(d[0], d[1], big.n_scans, big.n_points, d[2], d[3], st)     # ~0.35 ms of persistent workgroups each
        for s in range(batch.n_scans):
            Rs, ts, r = icp.fit(*batch.scan(s), batch.R[s], batch.t[s], 5.0)
            assert r.iters == iters[s] and r.n_corr == ncorr[s]
            assert np.abs(ts - to[s]).max() < POS_TOL and ang_diff(yaw(Rs), yaw(Ro[s]))[0] < ANG_TOL
    st.synchronize()
    dt = time.perf_counter() - t0
    assert dt < 1.0, "24 fits beside 32 raycasts took %.3f s" % dt
    icp.close()
    g.close()


# ---- the tile form (round 6): a model that does not fit LDS, scene points dealt over the workgroups along a Morton curve,
# their state in LDS slots, the tile of the index they can reach staged into LDS (icp_single.hip)

def one_scan_batch(s_ga, s_nga, R0, t0):
    pts = np.ascontiguousarray(np.concatenate([np.asarray(s_ga, np.float64).reshape(-1, 2), np.asarray(s_nga, np.float64).reshape(-1, 2)]))
    return synth.ScanBatch(pts, np.array([0, len(pts)], np.int32), np.array([len(s_ga)], np.int32),
                           np.asarray(R0, np.float64).reshape(1, 4), np.asarray(t0, np.float64).reshape(1, 2), np.zeros((1, 3)))


def test_tile_form_on_config3_matches():
    """Three matches of config 3 (tests/golden/spread_case3.npz: 19 999 + 871 model points, cells of hundreds of stacked wall
    points; scenes of ~600 voxel centroids in the filter's order): oracle, tile form, and the form without tiles step by step."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spread_case3.npz"))
    m_ga, m_nga = d["m_ga"].astype(np.float64), d["m_nga"].astype(np.float64)
    model = O.IcpModel(m_ga, m_nga)
    icp = api.Icp(m_ga, m_nga)
    plain = api.Icp(m_ga, m_nga, spread_tile=-1)
    assert not icp.index_info()["in_lds"]
    for k in (1, 5, 9):
        batch = one_scan_batch(d["s_ga%d" % k], d["s_nga%d" % k], d["R%d" % k], d["t%d" % k])
        Ro, to, otr, steps = model.fit(d["s_ga%d" % k], d["s_nga%d" % k], d["R%d" % k].reshape(2, 2), d["t%d" % k], O.icp_params(20, 1e-6, 5.0))
        R, t, res, tr = icp.fit_batch(batch, trace=True)
        Rp, tp, resp, trp = plain.fit_batch(batch, trace=True)
        assert res["iters"][0] == steps == resp["iters"][0]
        for it in range(steps):                                   # the same correspondences in every step
            assert tr[0, it, 7] == otr[it, 7] == trp[0, it, 7], (k, it)
            assert np.abs(tr[0, it, :6] - otr[it, :6]).max() < 1e-9 and np.abs(tr[0, it, :6] - trp[0, it, :6]).max() < 1e-9, (k, it)
        R2, t2, res2, _ = icp.fit_batch(batch)
        assert np.array_equal(R, R2) and np.array_equal(t, t2)   # bitwise reproducible
    icp.close()
    plain.close()


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("mode", ["p2p", "p2l"])
def test_tile_form_far_starts_and_many_scans(mode, form):
    """Starts 0.6 m / 0.1 rad off (queries leave their first tiles: searches through L2 and stagings until the budget is spent),
    1, 3 and 16 scans per launch (256, 85, 16 workgroups each: one to five passes of slots per workgroup), both solvers, against
    the oracle and the form without tiles."""
    m_ga, m_nga = synth.make_map(39998)
    kw = dict(mode=api.ICP_P2L, normals_k=10) if mode == "p2l" else {}
    omode = (O.NN_KDTREE, O.MODE_P2L) if mode == "p2l" else (O.NN_KDTREE,)
    model = O.IcpModel(m_ga, m_nga, **({"normals_k": 10} if mode == "p2l" else {}))
    icp = api.Icp(m_ga, m_nga, max_iter=25, min_delta=1e-6, spread_tile=form, **kw)   # 1: tiles staged into LDS, 2: the index where it lies
    plain = api.Icp(m_ga, m_nga, max_iter=25, min_delta=1e-6, spread_tile=-1, **kw)
    for n_scans in (1, 3, 16):
        batch = synth.make_batch(n_scans, n_loop=256)
        rs = np.random.RandomState(n_scans)
        for s in range(n_scans):
            x, y, th = batch.true_poses[s]
            R0, t0 = synth.pose_to_Rt(x + rs.uniform(-0.6, 0.6), y + rs.uniform(-0.6, 0.6), th + rs.uniform(-0.1, 0.1))
            batch.R[s], batch.t[s] = R0.reshape(4), t0
        Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, O.icp_params(25, 1e-6, 5.0, *omode))
        R, t, res, _ = icp.fit_batch(batch)
        Rp, tp, resp, _ = plain.fit_batch(batch)
        assert np.array_equal(res["iters"], resp["iters"]) and np.array_equal(res["n_corr"], resp["n_corr"])
        assert np.abs(t - tp).max() < 1e-9 and np.abs(R - Rp).max() < 1e-9
        assert np.array_equal(res["iters"], iters) and np.array_equal(res["n_corr"], ncorr)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    icp.close()
    plain.close()


@pytest.mark.parametrize("form", [1, 2])
def test_tile_form_ties_and_ragged_scans(form):
    """The edge cases of test_spread_edge_cases with the index kept in HBM/L2 (the tile form): exact distance ties on a gridded
    model with duplicates leave the tile for the exact search (lowest original index, the brute-force arbiter's rule), scans
    below 5 points, without correspondences, with a skipped class; plus a scan of 2049 points (beyond the tile form: the same
    launch serves it without tiles)."""
    rs = np.random.RandomState(11)
    gx, gy = np.meshgrid(np.arange(60) * 0.5, np.arange(40) * 0.5)
    grid = np.stack([gx.ravel(), gy.ravel()], 1)
    m_nga = np.concatenate([grid, grid[:200]])
    m_ga = grid[:3] + 0.1
    R0, t0 = synth.pose_to_Rt(0.1, -0.05, 0.01)
    scans = [grid[rs.choice(len(grid), 500)] + 0.25, np.zeros((3, 2)), np.full((10, 2), 900.0),
             grid[rs.choice(len(grid), 64)] + rs.randn(64, 2) * 0.02,
             grid[rs.choice(len(grid), 2049)] + rs.randn(2049, 2) * 0.02,
             grid[rs.choice(len(grid), 700)] + 0.25 + rs.randn(700, 2) * 1e-3]   # near-ties: decided on the tile
    nga = [0, 1, 0, 10, 0, 0]
    off = np.cumsum([0] + [len(x) for x in scans]).astype(np.int32)
    Rs, ts = np.tile(R0.reshape(4), (6, 1)), np.tile(t0, (6, 1))
    Rs[0], ts[0] = [1, 0, 0, 1], [0, 0]
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(scans)), off, np.array(nga, np.int32), Rs, ts, np.zeros((6, 3)))
    icp, R, t, res, _ = check_against_oracle(m_ga, m_nga, batch, 15, 1e-6, nn=O.NN_BRUTE, force_global=1, spread_tile=form)
    assert not icp.index_info()["in_lds"]
    assert res["iters"][1] == 0 and np.array_equal(R[1], R0.reshape(4)) and np.array_equal(t[1], t0)
    assert (res["iters"][2], res["n_corr"][2], res["delta"][2]) == (1, 0, -1.0)
    assert res["n_corr"][3] == 54
    icp.close()


def _random_model(rs, n_total, with_duplicates):
    """Model shapes the index meets: blobs of hundreds of points inside one lattice cell (stacked lidar returns), walls (lines with
    centimetre noise), thin scatter, exact duplicates; points tagged GA / NGA at random with one class possibly small."""
    parts = []
    n_blob = n_total // 3
    for _ in range(rs.randint(6, 30)):
        c = rs.uniform(-18, 18, 2)
        parts.append(c + rs.normal(0, rs.choice([0.005, 0.02, 0.08]), (n_blob // 12, 2)))
    n_wall = n_total // 3
    for _ in range(8):
        a, b = rs.uniform(-20, 20, 2), rs.uniform(-20, 20, 2)
        u = rs.uniform(0, 1, (n_wall // 8, 1))
        parts.append(a + u * (b - a) + rs.normal(0, 0.01, (n_wall // 8, 2)))
    pts = np.concatenate(parts)
    rest = max(n_total - len(pts), 16)
    pts = np.concatenate([pts, rs.uniform(-22, 22, (rest, 2))])
    if with_duplicates:
        pts[rs.randint(0, len(pts), 200)] = pts[rs.randint(0, len(pts), 200)]
    pts = pts[rs.permutation(len(pts))][:n_total]
    n_ga = int(len(pts) * rs.choice([0.03, 0.3, 0.5]))
    return np.ascontiguousarray(pts[:n_ga]), np.ascontiguousarray(pts[n_ga:])


def _random_scans(rs, m_ga, m_nga, n_scans):
    """Scans = model points seen from a pose 0.05-0.8 m / up to 0.12 rad off, with noise, a share of points the model never saw
    (inside and far outside its lattice), ragged sizes."""
    pts, off, nga, Rs, ts = [], [0], [], [], []
    for s in range(n_scans):
        n = int(rs.choice([7, 60, 400, 1100, 1900]))
        k_ga = min(int(n * rs.uniform(0.05, 0.6)), len(m_ga))
        ga = m_ga[rs.randint(0, len(m_ga), k_ga)] + rs.normal(0, 0.01, (k_ga, 2))
        ng = m_nga[rs.randint(0, len(m_nga), n - k_ga)] + rs.normal(0, 0.01, (n - k_ga, 2))
        for a in (ga, ng):                                        # outliers: unseen parts, and points far outside the lattice
            k = len(a) // 5
            if k:
                a[:k] = rs.uniform(-30, 30, (k, 2))
            if len(a) > 3:
                a[-1] = rs.uniform(200, 400, 2)
        th, tx, ty = rs.uniform(-0.12, 0.12), rs.uniform(-0.8, 0.8), rs.uniform(-0.8, 0.8)
        c, sn = np.cos(th), np.sin(th)
        Rt = np.array([[c, -sn], [sn, c]])
        # the scan in its own frame: model = R p + t  ->  p = R^T (model - t); the fit starts from the identity
        ga, ng = (ga - [tx, ty]) @ Rt, (ng - [tx, ty]) @ Rt
        pts += [ga, ng]
        off.append(off[-1] + n)
        nga.append(k_ga)
        Rs.append([1.0, 0.0, 0.0, 1.0])
        ts.append([0.0, 0.0])
    return synth.ScanBatch(np.ascontiguousarray(np.concatenate(pts)), np.array(off, np.int32), np.array(nga, np.int32),
                           np.array(Rs, np.float64), np.array(ts, np.float64), np.zeros((n_scans, 3)))


@pytest.mark.parametrize("seed", range(6))
def test_spread_forms_on_random_models(seed):
    """Seeded random models (dense blobs, walls, scatter, duplicates; 6 k - 36 k points: index in LDS, just above it, at the cap) and
    ragged scans with outliers: every spread form -- tiles staged into LDS, the index where it lies, round 5's form -- against
    the oracle: iteration counts and correspondences equal, poses to the tolerance, the forms among themselves to 1e-9."""
    rs = np.random.RandomState(1000 + seed)
    n_total = [6000, 21000, 36000, 14000, 25000, 30000][seed]
    m_ga, m_nga = _random_model(rs, n_total, with_duplicates=seed % 2 == 0)
    model = O.IcpModel(m_ga, m_nga)
    n_scans = [2, 2, 5, 4, 3, 16][seed]
    batch = _random_scans(rs, m_ga, m_nga, n_scans)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, O.icp_params(20, 1e-6, 5.0, O.NN_KDTREE))
    first = None
    for form in (1, 2, -1):
        icp = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6, spread_tile=form)
        R, t, res, _ = icp.fit_batch(batch)
        icp.close()
        assert np.array_equal(res["iters"], iters), (form, res["iters"], iters)
        assert np.array_equal(res["n_corr"], ncorr), (form, res["n_corr"], ncorr)
        assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL, form
        if first is None:
            first = (R, t)
        else:
            assert np.abs(t - first[1]).max() < 1e-9 and np.abs(R - first[0]).max() < 1e-9, form


@pytest.mark.parametrize("big", [False, True])
def test_spread_fit_replayed_as_a_graph(big):
    """One scan's fit captured into a hipGraph (slam_graph_begin_capture) and replayed with other initial poses in the device
    arrays it reads: every replay is a fit of its own (the launch's granule tags are the host's at capture time: a replay takes a
    fill and a tag range of its own into the graph instead of meeting its predecessor's granules)."""
    m_ga, m_nga = synth.make_map(39998 if big else 10000)
    model = O.IcpModel(m_ga, m_nga)
    icp = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6)
    batch = synth.make_batch(4, n_loop=256)
    ga, nga = batch.scan(0)
    pts = np.ascontiguousarray(np.concatenate([ga, nga]))
    d_pts = api.DeviceArray.from_host(pts, np.float64)
    d_off = api.DeviceArray.from_host(np.array([0, len(pts)], np.int32))
    d_nga = api.DeviceArray.from_host(np.array([len(ga)], np.int32))
    d_R0, d_t0 = api.DeviceArray((1, 4), np.float64), api.DeviceArray((1, 2), np.float64)
    d_R, d_t = api.DeviceArray((1, 4), np.float64), api.DeviceArray((1, 2), np.float64)
    d_res = api.DeviceArray((1,), api.RESULT_DTYPE)
    st = api.Stream()
    x, y, th = batch.true_poses[0]
    starts = [synth.pose_to_Rt(x + dx, y + dy, th + dth) for dx, dy, dth in ((0.1, -0.1, 0.02), (-0.3, 0.2, -0.05), (0.05, 0.4, 0.08), (0.1, -0.1, 0.02))]
    d_R0.upload(np.asarray(starts[0][0], np.float64).reshape(1, 4)), d_t0.upload(np.asarray(starts[0][1], np.float64).reshape(1, 2))
    icp.fit_batch_from_dev(d_pts, d_off, d_nga, 1, d_R0, d_t0, d_R, d_t, 5.0, d_res, None, st)    # (buffers made outside the capture)
    st.synchronize()
    graph = api.Graph(st)
    with graph:
        icp.fit_batch_from_dev(d_pts, d_off, d_nga, 1, d_R0, d_t0, d_R, d_t, 5.0, d_res, None, st)
    for R0, t0 in starts:
        d_R0.upload(np.asarray(R0, np.float64).reshape(1, 4)), d_t0.upload(np.asarray(t0, np.float64).reshape(1, 2))
        graph.launch(st)
        st.synchronize()
        Ro, to, tro, steps = model.fit(ga, nga, np.asarray(R0).reshape(2, 2), np.asarray(t0), O.icp_params(20, 1e-6, 5.0))
        res = d_res.download()[0]
        assert res["iters"] == steps and res["n_corr"] == int(tro[steps - 1, 7])
        assert np.abs(d_t.download()[0] - to).max() < POS_TOL and ang_diff(yaw(d_R.download()[0]), yaw(Ro)).max() < ANG_TOL
    # ... and an ordinary launch behind the replays does not take their granules for its own
    icp.fit_batch_from_dev(d_pts, d_off, d_nga, 1, d_R0, d_t0, d_R, d_t, 5.0, d_res, None, st)
    st.synchronize()
    assert np.abs(d_t.download()[0] - to).max() < POS_TOL
    icp.close()
