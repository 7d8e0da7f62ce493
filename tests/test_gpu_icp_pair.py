"""Two scans per workgroup (icp_fit_pair_kernel, slam_icp_params::pair_scans): each half of a workgroup runs the fused
ring-search / list-sweep schedule on a scan of its own against ONE LDS index, synchronising through a counter in LDS.
The results are the oracle's whatever the partner does: odd batches (an idle half), ragged sizes, scans that leave the
ring search early, late or never, scans that stop on min_delta while the partner runs on."""
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


@pytest.fixture(scope="module")
def world():
    m_ga, m_nga = synth.make_map()
    return m_ga, m_nga, O.IcpModel(m_ga, m_nga)


def check(world, batch, max_iter, min_delta, pair, indist=5.0):
    m_ga, m_nga, model = world
    icp = api.Icp(m_ga, m_nga, max_iter=max_iter, min_delta=min_delta, pair_scans=pair, spread_scans=-1)
    R, t, res, _ = icp.fit_batch(batch, indist=indist)
    R2, t2, res2, _ = icp.fit_batch(batch, indist=indist)
    icp.close()
    assert np.array_equal(R, R2) and np.array_equal(t, t2) and np.array_equal(res["n_corr"], res2["n_corr"])   # reproducible
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(max_iter, min_delta, indist))
    assert np.array_equal(res["iters"], iters), (res["iters"], iters)
    assert np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < POS_TOL and ang_diff(yaw(R), yaw(Ro)).max() < ANG_TOL
    assert np.abs(res["delta"] - delta).max() < 1e-9
    return R, t, res


@pytest.mark.parametrize("pair", [1, 2])
@pytest.mark.parametrize("n_scans", [33, 2])
def test_pairs_match_oracle(world, pair, n_scans):
    """an odd batch leaves the last workgroup one idle half"""
    check(world, synth.make_batch(n_scans, n_loop=256), 30, -1.0, pair)


@pytest.mark.parametrize("pair", [1, 2])
def test_pairs_ragged_scan_sizes(world, pair):
    """5 to ~2600 points: partners with very different pass counts, a scan shorter than one cooperative round"""
    pts, off, nga, Rs, ts = [], [0], [], [], []
    for k, beams in enumerate([6, 2600, 41, 64, 66, 700, 1026, 514, 1027, 1081, 1090, 513, 1100, 2200, 30]):
        ga, ng, pose = synth.make_scan(3 * k, 256, n_beams=beams)
        pts += [ga, ng]
        off.append(off[-1] + len(ga) + len(ng))
        nga.append(len(ga))
        R, t = synth.pose_to_Rt(*synth.init_pose(3 * k, pose))
        Rs.append(R.reshape(4))
        ts.append(t)
    batch = synth.ScanBatch(np.ascontiguousarray(np.concatenate(pts)), np.array(off, np.int32), np.array(nga, np.int32),
                            np.array(Rs), np.array(ts), np.zeros((len(nga), 3)))
    check(world, batch, 14, -1.0, pair)


@pytest.mark.parametrize("frac", [0.01, 0.5])
def test_pairs_with_outlier_scans(world, frac):
    """every second scan carries outliers: at 50 % it never leaves the ring search while its partner wants the lists"""
    batch = synth.make_batch(8, n_loop=256)
    rs = np.random.RandomState(int(frac * 1000))
    pts = batch.pts.copy()
    for s in range(0, batch.n_scans, 2):
        o, e = batch.scan_off[s], batch.scan_off[s + 1]
        k = rs.choice(np.arange(o, e), int(frac * (e - o)), replace=False)
        pts[k] += rs.uniform(-4.0, 4.0, (len(k), 2))
    noisy = synth.ScanBatch(pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, batch.true_poses)
    check(world, noisy, 25, 1e-7, 2)


def test_pairs_edge_iteration_counts(world):
    """max_iter below, at and above the hand-over iteration; min_delta reached in either form, partners stopping apart"""
    batch = synth.make_batch(6, n_loop=256)
    for max_iter, min_delta in ((1, -1.0), (9, -1.0), (10, -1.0), (11, -1.0), (40, 1e-2), (40, 1e-4), (40, 1e-9)):
        _, _, res = check(world, batch, max_iter, min_delta, 2)
        if min_delta >= 1e-4:
            assert (res["iters"] < 40).all()


def test_pairs_too_few_points_and_empty_class(world):
    """a partner with fewer than 5 points is left untouched (icp.cpp:100-103) while the other registers"""
    m_ga, m_nga, model = world
    big = synth.make_batch(3, n_loop=256)
    pts = [big.pts[big.scan_off[0]:big.scan_off[1]], np.array([[1.0, 2.0], [2.0, 1.0], [0.5, 0.5]]), big.pts[big.scan_off[2]:big.scan_off[3]]]
    off = np.array([0, len(pts[0]), len(pts[0]) + 3, len(pts[0]) + 3 + len(pts[2])], np.int32)
    nga = np.array([big.scan_nga[0], 1, big.scan_nga[2]], np.int32)
    batch = synth.ScanBatch(np.concatenate(pts), off, nga, big.R, big.t, big.true_poses)
    icp = api.Icp(m_ga, m_nga, max_iter=20, min_delta=1e-6, pair_scans=2, spread_scans=-1)
    R, t, res, _ = icp.fit_batch(batch)
    icp.close()
    assert np.array_equal(R[1], big.R[1]) and np.array_equal(t[1], big.t[1]) and res["iters"][1] == 0
    assert np.abs(t[[0, 2]] - big.true_poses[[0, 2], :2]).max() < 0.03


def test_default_pairs_from_two_scans_per_cu(world):
    """the library default: one scan per workgroup below two scans per CU, pairs from there on -- same results"""
    m_ga, m_nga, _ = world
    n_cu = api.device_info()[1]
    batch = synth.make_batch(2 * n_cu, n_loop=2 * n_cu)
    out = []
    for pair in (0, -1):
        icp = api.Icp(m_ga, m_nga, max_iter=12, min_delta=-1.0, pair_scans=pair)
        out.append(icp.fit_batch(batch))
        icp.close()
    assert np.array_equal(out[0][2]["n_corr"], out[1][2]["n_corr"])
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-8 and np.abs(out[0][0] - out[1][0]).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["fused", "pairs", "spread", "two launches", "ring only"])
def test_fit_from_separate_initial_poses_equals_the_fit_in_place(form):
    """slam_icp_fit_batch_from_dev: initial poses read from one pair of arrays, registered poses written to another -- the
    same bits as slam_icp_fit_batch_dev leaves in place, in every launch form; the initial poses are not written; a scan
    of fewer than five points comes back with its initial pose (icp.cpp:100-103)."""
    from slam_amd import api
    m_ga, m_nga = synth.make_map(10000)
    n_scans = {"spread": 3}.get(form, 12)
    batch = synth.make_batch(n_scans, n_loop=64)
    off = batch.scan_off.copy()
    # the last scan shrinks to four points
    pts = batch.pts[:off[-2] + 4].copy()
    off[-1] = off[-2] + 4
    nga = batch.scan_nga.copy()
    nga[-1] = min(nga[-1], 2)
    kw = {"fused": dict(pair_scans=-1), "pairs": dict(pair_scans=2), "spread": {}, "two launches": dict(split_launch=1),
          "ring only": dict(lanes_per_point=4)}[form]
    icp = api.Icp(m_ga, m_nga, max_iter=12, min_delta=-1.0, **kw)
    d_pts = api.DeviceArray.from_host(pts, np.float64)
    d_off = api.DeviceArray.from_host(off, np.int32)
    d_nga = api.DeviceArray.from_host(nga, np.int32)
    d_R0, d_t0 = api.DeviceArray.from_host(batch.R, np.float64), api.DeviceArray.from_host(batch.t, np.float64)
    d_Ra, d_ta = api.DeviceArray.from_host(batch.R, np.float64), api.DeviceArray.from_host(batch.t, np.float64)
    d_Rb, d_tb = api.DeviceArray(batch.R.shape, np.float64), api.DeviceArray(batch.t.shape, np.float64)
    d_Rb.zero(); d_tb.zero()
    ra, rb = api.DeviceArray((n_scans,), api.RESULT_DTYPE), api.DeviceArray((n_scans,), api.RESULT_DTYPE)
    icp.fit_batch_dev(d_pts, d_off, d_nga, n_scans, d_Ra, d_ta, 5.0, ra)
    icp.fit_batch_from_dev(d_pts, d_off, d_nga, n_scans, d_R0, d_t0, d_Rb, d_tb, 5.0, rb)
    api.synchronize()
    Ra, ta, Rb, tb = d_Ra.download(), d_ta.download(), d_Rb.download(), d_tb.download()
    assert np.array_equal(Ra, Rb) and np.array_equal(ta, tb)
    assert np.array_equal(ra.download()["iters"], rb.download()["iters"]) and (rb.download()["iters"][:-1] == 12).all()
    assert np.array_equal(d_R0.download(), batch.R) and np.array_equal(d_t0.download(), batch.t)
    assert np.array_equal(Rb[-1], batch.R[-1]) and np.array_equal(tb[-1], batch.t[-1]) and rb.download()["iters"][-1] == 0
    assert np.abs(tb[:-1] - batch.true_poses[:-1, :2]).max() < 0.3
    icp.close()
