"""BASELINE configs 2 and 4 at FULL size against the oracle, no sampling: every scan of the batch through
oicp_fit_batch (OpenMP over scans), every beam of the batch through the sequential-per-beam Bresenham oracle
(ogrid_raycast_mt: OpenMP over beams, atomic increments -- integer sums, order-free).  Tolerances as everywhere:
correspondence counts equal, poses within 1e-4 m / 1e-5 rad, grid counts bit-exact."""
import os

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth

pytestmark = pytest.mark.gpu
THREADS = max(1, min(os.cpu_count() or 1, 32))


def yaw(R):
    R = np.asarray(R).reshape(-1, 4)
    return np.arctan2(R[:, 2], R[:, 0])


def ang_diff(a, b):
    d = a - b
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


@pytest.mark.parametrize("n_scans,size,min_delta", [(256, 2000, -1.0), (256, 2000, 1e-6), (1024, 4000, -1.0)],
                         ids=["config2-fixed-30", "config2-early-exit", "config4-share"])
def test_full_batch_matches_oracle(n_scans, size, min_delta):
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(n_scans)
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=min_delta)
    R, t, res, _ = icp.fit_batch(batch, indist=5.0)
    icp.close()
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(30, min_delta, 5.0), n_threads=THREADS)
    assert np.array_equal(res["iters"], iters)
    assert np.array_equal(res["n_corr"], ncorr)
    assert np.abs(t - to).max() < 1e-4 and ang_diff(yaw(R), yaw(Ro)).max() < 1e-5
    # the last step's size: the sums run in another order, and over a thousand scans one query rounds to the other
    # side of a float somewhere, which moves that scan's last step by parts in a million (the poses stay within tolerance)
    assert np.abs(res["delta"] - delta).max() < 1e-5

    # the whole batch ray-cast from the GPU's own poses, every implementation, against the oracle
    gp = O.grid_params(size, size, 0.05, min_cluster_points=20)
    ends, origins = [], []
    for s in range(n_scans):
        o, e = batch.scan_off[s], batch.scan_off[s + 1]
        ends.append(O.transform_points(batch.pts[o:e], R[s], t[s]))
        origins.append(np.tile(t[s].astype(np.float32), (e - o, 1)))
    H, M, upd = O.grid_raycast(gp, np.concatenate(origins), np.concatenate(ends), n_threads=THREADS)
    d = [api.DeviceArray.from_host(a, dt) for a, dt in
         ((batch.pts, np.float64), (batch.scan_off, np.int32), (R, np.float64), (t, np.float64))]
    for impl in (api.RAYCAST_TILED, api.RAYCAST_TILED_MERGE):
        g = api.Grid(size, size, 0.05, rolling=0, min_cluster_points=20, raycast_impl=impl)
        g.raycast_scans_dev(d[0], d[1], n_scans, batch.n_points, d[2], d[3])
        g.finalize()
        api.synchronize()
        hits, misses = g.read_counts()
        assert np.array_equal(hits, H) and np.array_equal(misses, M), impl
        assert g.total_updates() == upd == int(H.sum()) + int(M.sum())
        num, occ = np.zeros(size * size), np.full(size * size, -1, np.int8)
        O.grid_finalize(gp, H, M, num, occ)
        assert np.array_equal(g.read_occupancy(), occ)
        assert np.array_equal(g.read_num_pts(), num)
        g.close()


@pytest.mark.parametrize("pair,min_delta", [(0, -1.0), (2, -1.0), (0, 1e-6)], ids=["one-per-workgroup", "pairs", "early-exit"])
def test_point_to_line_config2_matches_own_oracle(pair, min_delta):
    """The solver north_star names -- point-to-line error, 3x3 normal equations per iteration
    (icpPointToPlane.cpp:37-107, normals :279-305; stale upstream, so the oracle is the build's own restatement,
    oracle/slam_oracle.c fit_step_p2l) -- on BASELINE config 2: all 256 scans x 30 iterations against the 10 k-point
    map, in the two batch forms the pipelined step uses, delta of EVERY step of EVERY scan against the oracle's trace."""
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(256)
    icp = api.Icp(m_ga, m_nga, mode=api.ICP_P2L, normals_k=10, max_iter=30, min_delta=min_delta, pair_scans=pair)
    assert icp.index_info()["two_forms"]
    R, t, res, trace = icp.fit_batch(batch, trace=True)
    icp.close()
    model = O.IcpModel(m_ga, m_nga, normals_k=10)
    prm = O.icp_params(30, min_delta, 5.0, O.NN_KDTREE, O.MODE_P2L)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, prm, n_threads=THREADS)
    assert np.array_equal(res["iters"], iters)
    assert np.array_equal(res["n_corr"], ncorr) and np.array_equal(ncorr, np.diff(batch.scan_off))
    assert np.abs(t - to).max() < 1e-4 and ang_diff(yaw(R), yaw(Ro)).max() < 1e-5
    assert np.abs(res["delta"] - delta).max() < 1e-5
    if min_delta < 0:
        assert (iters == 30).all()
        worst, ties = 0.0, []
        for s in range(256):
            t_ga, t_nga = batch.scan(s)
            _, _, tr, steps = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], prm)
            assert steps == 30
            err = max(np.abs(trace[s, :, 6] - tr[:, 6]).max(), np.abs(trace[s, :, :6] - tr[:, :6]).max())
            if err >= 1e-7:
                # an exact float distance tie somewhere in the scan's 32 k searches: the kd-tree takes the candidate it visits
                # last (kdtree.cpp:612-618), the GPU the lowest index, as the brute-force arbiter does (kdtree.cpp:360-375)
                ties.append(s)
                _, _, tr, _ = model.fit(t_ga, t_nga, batch.R[s], batch.t[s], O.icp_params(30, min_delta, 5.0, O.NN_BRUTE, O.MODE_P2L))
                err = max(np.abs(trace[s, :, 6] - tr[:, 6]).max(), np.abs(trace[s, :, :6] - tr[:, :6]).max())
            worst = max(worst, err)
        assert worst < 1e-7, (worst, ties)   # the pose and the step size after every one of the 30 steps of every scan
        assert len(ties) <= 16, ties
    else:
        # (a few scans never get below 1e-6: a query between two model points changes sides step after step -- in the oracle too)
        assert np.median(iters) < 20 and (iters < 30).sum() > 200
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.03


def test_config5_stream_of_10240_scans_matches_oracle():
    """BASELINE config 5 at full length through slam_mapper_*: 40 chunks of 256 scans from pinned host memory, two
    registration streams with two scans per workgroup, five chunks in flight, finalize every 8 chunks -- with the prior
    map as a fixed target, so that every scan has ONE right answer: the oracle's registration of every scan (poses within
    tolerance, correspondence-exact by the same tests at batch size) and the oracle's Bresenham of all 11 M beams from
    the mapper's own poses (counts bit-exact).  (The sliding-window form takes whatever chunks have finished when a
    rebuild looks -- timing-dependent by design -- and is held against the oracle with strict_window in
    tests/test_gpu_mapper.py.)"""
    n_scans, chunk, size = 10240, 256, 2000
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(n_scans, n_loop=n_scans)
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=size, grid_size_y=size, resolution=0.05,
                    max_scans=chunk, max_points=chunk * 1100, icp=dict(max_iter=30, min_delta=-1.0), merge_every=8)
    assert mp.n_slots == 5
    R, t = np.zeros((n_scans, 4)), np.zeros((n_scans, 2))
    pending = []
    for k in range(n_scans // chunk):
        if len(pending) == mp.n_slots:
            slot, a = pending.pop(0)
            R[a:a + chunk], t[a:a + chunk] = mp.wait(slot)
        pending.append((mp.push(batch.shard(k, n_scans // chunk)), k * chunk))
    for slot, a in pending:
        R[a:a + chunk], t[a:a + chunk] = mp.wait(slot)
    mp.finish()
    hits, misses = mp.grid.read_counts()
    occ = mp.grid.read_occupancy()
    assert mp.stats()["chunks"] == n_scans // chunk
    mp.close()
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(30, -1.0, 5.0), n_threads=THREADS)
    assert np.abs(t - to).max() < 1e-4 and ang_diff(yaw(R), yaw(Ro)).max() < 1e-5
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.03
    gp = O.grid_params(size, size, 0.05, min_cluster_points=20)
    ends, origins = [], []
    for s in range(n_scans):
        o, e = batch.scan_off[s], batch.scan_off[s + 1]
        ends.append(O.transform_points(batch.pts[o:e], R[s], t[s]))
        origins.append(np.tile(t[s].astype(np.float32), (e - o, 1)))
    H, M, upd = O.grid_raycast(gp, np.concatenate(origins), np.concatenate(ends), n_threads=THREADS)
    assert np.array_equal(hits, H) and np.array_equal(misses, M)
    num, eocc = np.zeros(size * size), np.full(size * size, -1, np.int8)
    O.grid_finalize(gp, H, M, num, eocc)
    assert np.array_equal(occ, eocc)


def test_config5_sliding_window_at_full_length_matches_the_oracle_schedule():
    """BASELINE config 5 in its own form, at full length: 10 240 scans in 40 chunks through slam_mapper_* against a
    SLIDING-WINDOW target -- the 5 k-point prior map plus at most 5 k points of the last four registered chunks thinned at
    0.1 m, rebuilt on the device every four chunks -- with strict_window, so that the schedule is the one the oracle can
    follow: the target of chunks 4r .. 4r + 3 is built from the oracle's OWN registrations of chunks 4r - 4 .. 4r - 1.
    Every pose within the north-star tolerance of the oracle's, all 11 M beams' counts and the occupancy bit-exact."""
    from test_gpu_mapper import thin_points
    n_scans, chunk, size, res, W, every, target_points, thin = 10240, 256, 2000, 0.05, 4, 4, 5000, 0.1
    m_ga, m_nga = synth.make_map(5000)
    batch = synth.make_batch(n_scans, n_loop=n_scans)
    n_chunks = n_scans // chunk
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=size, grid_size_y=size, resolution=res,
                    max_scans=chunk, max_points=chunk * 1100, icp=dict(max_iter=30, min_delta=-1.0), window_chunks=W,
                    rebuild_every=every, keep_prior=1, target_points=target_points, thin_res=thin, merge_every=8, strict_window=1)
    R, t = np.zeros((n_scans, 4)), np.zeros((n_scans, 2))
    pending = []
    for k in range(n_chunks):
        if len(pending) == mp.n_slots:
            slot, a = pending.pop(0)
            R[a:a + chunk], t[a:a + chunk] = mp.wait(slot)
        pending.append((mp.push(batch.shard(k, n_chunks)), k * chunk))
    for slot, a in pending:
        R[a:a + chunk], t[a:a + chunk] = mp.wait(slot)
    mp.finish()
    hits, misses = mp.grid.read_counts()
    occ = mp.grid.read_occupancy()
    st = mp.stats()
    mp.close()
    assert st["chunks"] == n_chunks and st["rebuilds"] == n_chunks // every - 1      # before chunks 4, 8, ..., 36

    # the oracle on the same schedule, with its own registrations in the window
    Ro, to = np.zeros((n_scans, 4)), np.zeros((n_scans, 2))
    window, model = [], O.IcpModel(m_ga, m_nga)
    for k in range(n_chunks):
        if k > 0 and k % every == 0:
            ga = np.concatenate([m_ga, thin_points(np.concatenate([w[0] for w in window[-W:]]), thin, size * res, target_points // 2)])
            nga = np.concatenate([m_nga, thin_points(np.concatenate([w[1] for w in window[-W:]]), thin, size * res, target_points // 2)])
            model = O.IcpModel(ga, nga)
        c = batch.shard(k, n_chunks)
        Rc, tc, _, _, _ = model.fit_batch(c.pts, c.scan_off, c.scan_nga, c.R, c.t, O.icp_params(30, -1.0, 5.0), n_threads=THREADS)
        Ro[k * chunk:(k + 1) * chunk], to[k * chunk:(k + 1) * chunk] = Rc, tc
        reg_ga, reg_nga = [], []
        for s in range(c.n_scans):
            p = c.pts[c.scan_off[s]:c.scan_off[s + 1]]
            q = np.stack([(Rc[s, 0] * p[:, 0] + Rc[s, 1] * p[:, 1]) + tc[s, 0], (Rc[s, 2] * p[:, 0] + Rc[s, 3] * p[:, 1]) + tc[s, 1]], 1)
            reg_ga.append(q[:c.scan_nga[s]])
            reg_nga.append(q[c.scan_nga[s]:])
        window.append((np.concatenate(reg_ga), np.concatenate(reg_nga)))
    assert np.abs(t - to).max() < 1e-4 and ang_diff(yaw(R), yaw(Ro)).max() < 1e-5
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.05

    gp = O.grid_params(size, size, res, min_cluster_points=20)
    ends, origins = [], []
    for s in range(n_scans):
        o, e = batch.scan_off[s], batch.scan_off[s + 1]
        ends.append(O.transform_points(batch.pts[o:e], R[s], t[s]))
        origins.append(np.tile(t[s].astype(np.float32), (e - o, 1)))
    H, M, upd = O.grid_raycast(gp, np.concatenate(origins), np.concatenate(ends), n_threads=THREADS)
    assert np.array_equal(hits, H) and np.array_equal(misses, M)
    num, eocc = np.zeros(size * size), np.full(size * size, -1, np.int8)
    O.grid_finalize(gp, H, M, num, eocc)
    assert np.array_equal(occ, eocc)


def test_endpoint_update_at_full_size_matches_oracle():
    """The grid update the reference actually performs -- MLS::addToOccupancy's endpoint loops (mls.cpp:73-142) -- on the
    BASELINE inputs at 2000 x 2000 @ 0.05 m: (a) config 2: every registered endpoint of the 256 scans as an obstacle point,
    one call; (b) config 3: ten 64-ring clouds, each split into its drv and ground sets by the ground segmentation, added
    scan by scan -- counts, evidence doubles and occupancy bit for bit, in the counting mode (counts summed, folded once) and
    in the in-order mode (the reference's sequential += 1.0 / -= 0.3 on the double, thresholds as it goes)."""
    size, res = 2000, 0.05
    gp = O.grid_params(size, size, res, min_cluster_points=20)
    # (a)
    m_ga, m_nga = synth.make_map()
    batch = synth.make_batch(256)
    icp = api.Icp(m_ga, m_nga, max_iter=30, min_delta=-1.0)
    R, t, res_, _ = icp.fit_batch(batch, indist=5.0)
    icp.close()
    ends = np.concatenate([O.transform_points(batch.pts[batch.scan_off[s]:batch.scan_off[s + 1]], R[s], t[s]) for s in range(256)])
    assert len(ends) == batch.n_points
    H, M = np.zeros(size * size, np.int32), np.zeros(size * size, np.int32)
    O.grid_add_endpoints(gp, ends, np.zeros((0, 2), np.float32), H, M)
    g = api.Grid(size, size, res, rolling=0, min_cluster_points=20)
    g.add_endpoints(ends, np.zeros((0, 2), np.float32))
    g.finalize()
    api.synchronize()
    hits, misses = g.read_counts()
    assert np.array_equal(hits, H) and np.array_equal(misses, M) and int(H.sum()) == batch.n_points and M.sum() == 0
    assert g.total_updates() == batch.n_points
    num, occ = np.zeros(size * size), np.full(size * size, -1, np.int8)
    O.grid_finalize(gp, H, M, num, occ)
    assert np.array_equal(g.read_occupancy(), occ) and np.array_equal(g.read_num_pts(), num)
    assert (occ == 100).sum() > 1000          # walls seen twenty times and more
    g.close()
    # (b)
    g = api.Grid(size, size, res, rolling=0, min_cluster_points=20)
    g_in = api.Grid(size, size, res, rolling=0, min_cluster_points=20)
    H[:], M[:] = 0, 0
    num_in, drv_in, occ_in = np.zeros(size * size), np.full(size * size, -1, np.int8), np.full(size * size, -1, np.int8)
    total = 0
    for k in range(10):
        xyz = synth.make_cloud3d(5 * k, n_loop=50)[0]
        lab = O.gseg_segment(xyz)[0]
        obs, gnd = np.ascontiguousarray(xyz[lab == O.GSEG_OBSTACLE]), np.ascontiguousarray(xyz[lab == O.GSEG_GROUND])
        assert len(obs) > 10000 and len(gnd) > 10000
        g.add_endpoints(obs, gnd)
        g_in.add_scan_inorder(obs, gnd)
        O.grid_add_endpoints(gp, obs, gnd, H, M)
        O.grid_add_scan_inorder(gp, obs, gnd, num_in, drv_in, occ_in)
        total += len(obs) + len(gnd)
    g.finalize()
    api.synchronize()
    hits, misses = g.read_counts()
    assert np.array_equal(hits, H) and np.array_equal(misses, M)
    assert g.total_updates() == int(H.sum()) + int(M.sum()) <= total
    num, occ = np.zeros(size * size), np.full(size * size, -1, np.int8)
    O.grid_finalize(gp, H, M, num, occ)
    assert np.array_equal(g.read_occupancy(), occ) and np.array_equal(g.read_num_pts(), num)
    assert np.array_equal(g_in.read_occupancy(), occ_in) and np.array_equal(g_in.read_num_pts(), num_in)   # bit-exact doubles
    assert (occ_in == 100).sum() > 0 and (occ_in == 0).sum() > 0
    g.close()
    g_in.close()
