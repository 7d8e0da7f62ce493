"""GPU parity of the occupancy-grid path through the C-ABI against the CPU
oracle (mls.cpp:59-150 restated; Bresenham = the build's own definition).
Cell indices and hit/miss counts are integers: the bar is bit-exact."""
import os

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth

pytestmark = pytest.mark.gpu


def og(grid, **kw):
    p = grid.params
    return O.grid_params(grid.size_x, grid.size_y, grid.resolution, p.max_range,
                         p.occupancy_increment, p.occupancy_decrement, p.min_cluster_points,
                         p.rolling, *grid.get_pose())


def test_endpoints_bit_exact_with_edge_values():
    rs = np.random.RandomState(1)
    g = api.Grid(500, 500, 0.1, min_cluster_points=20, rolling=1)
    obs = (rs.randn(20000, 4) * 12).astype(np.float32)       # PointXYZGD stride 4
    gnd = (rs.randn(30000, 4) * 12).astype(np.float32)
    # truncation toward zero, bounds, NaN / huge, the range gate
    obs[:8, :2] = [[-24.95, 0], [-25.05, 0], [-25.1, 0], [24.99, 24.99], [25.0, 0], [np.nan, 1],
                   [1e30, 0], [0, -3e9]]
    g.add_endpoints(obs, gnd)
    hits, misses = g.read_counts()
    eh, em, cells, n = O.grid_add_endpoints(og(g), obs, gnd)
    assert np.array_equal(hits, eh) and np.array_equal(misses, em)
    assert g.total_updates() == n
    g.close()


def test_range_gate_boundaries_rolling_and_global():
    # mls.cpp:82 float sqrt (rolling) vs :84-86 double sqrt from the pose (global frame)
    for rolling, pose in ((1, (0.0, 0.0)), (0, (50.0, -10.0))):
        g = api.Grid(4000, 4000, 0.05, rolling=rolling)
        if not rolling:
            g.set_pose(*pose)
        base = np.float32(75.0)
        xs = [base]
        for _ in range(40):
            xs.append(np.nextafter(xs[-1], np.float32(100)))
        for _ in range(40):
            xs.insert(0, np.nextafter(xs[0], np.float32(0)))
        rs = np.random.RandomState(3)
        ang = rs.uniform(0, 2 * np.pi, 4000)
        rad = np.array(rs.choice(xs, 4000), dtype=np.float64) + rs.randn(4000) * 1e-5
        pts = np.stack([pose[0] + rad * np.cos(ang), pose[1] + rad * np.sin(ang)], 1).astype(np.float32)
        axis = np.stack([pose[0] + np.array(xs, np.float64), np.full(len(xs), pose[1])], 1).astype(np.float32)
        pts = np.concatenate([pts, axis])
        g.add_endpoints(pts, np.zeros((0, 2), np.float32))
        hits, _ = g.read_counts()
        eh, _, cells, n = O.grid_add_endpoints(og(g), pts, np.zeros((0, 2), np.float32))
        assert np.array_equal(hits, eh)
        assert 0 < n < len(pts)       # the set really straddles the gate
        g.close()


def test_y_bound_quirk_non_square():
    # mls.cpp:90 tests y against size_x
    rs = np.random.RandomState(5)
    pts = (rs.rand(5000, 2) * 40 - 20).astype(np.float32)
    for sx, sy in ((100, 200), (200, 100)):
        g = api.Grid(sx, sy, 0.2, rolling=1)
        g.add_endpoints(pts, pts[:100])
        hits, misses = g.read_counts()
        eh, em, _, _ = O.grid_add_endpoints(og(g), pts, pts[:100])
        assert np.array_equal(hits, eh) and np.array_equal(misses, em)
        g.close()


@pytest.mark.parametrize("impl", [api.RAYCAST_TILED, api.RAYCAST_TILED_MERGE, api.RAYCAST_GLOBAL])
def test_raycast_config1_golden_and_oracle(impl, golden_dir):
    """BASELINE config 1 grid: 500 x 500 @ 0.1 m, one scan from its true pose."""
    G = np.load(os.path.join(golden_dir, "grid_golden.npz"))
    batch = synth.make_batch(1, n_loop=256)
    R, t = synth.pose_to_Rt(*G["true_pose"])
    end = O.transform_points(batch.pts, R, t)
    origin = np.tile(np.array(t, dtype=np.float32), (len(end), 1))
    g = api.Grid(500, 500, 0.1, min_cluster_points=20, rolling=0, raycast_impl=impl)
    g.raycast(origin, end)
    hits, misses = g.read_counts()
    assert np.array_equal(np.flatnonzero(hits), G["ray_hit_cells"])
    assert np.array_equal(hits[G["ray_hit_cells"]], G["ray_hit_counts"])
    assert np.array_equal(np.flatnonzero(misses), G["ray_miss_cells"])
    assert np.array_equal(misses[G["ray_miss_cells"]], G["ray_miss_counts"])
    assert g.total_updates() == int(G["ray_n_upd"])
    g.close()


@pytest.mark.parametrize("impl", [api.RAYCAST_TILED, api.RAYCAST_TILED_MERGE, api.RAYCAST_GLOBAL])
@pytest.mark.parametrize("size,res", [((300, 300), 0.25), ((257, 131), 0.3), ((64, 64), 1.0)])
def test_raycast_random_beams_bit_exact(impl, size, res):
    """All octants, beams leaving the window, degenerate beams, tile borders."""
    rs = np.random.RandomState(size[0])
    sx, sy = size
    n = 30000
    half = np.array([sx, sy]) * res / 2
    origin = (rs.rand(n, 2) * 2 - 1) * half * 1.1
    end = (rs.rand(n, 2) * 2 - 1) * half * 1.1
    end[:200] = origin[:200]                               # zero-length
    end[200:400, 1] = origin[200:400, 1]                   # horizontal
    end[400:600, 0] = origin[400:600, 0]                   # vertical
    d = rs.rand(200, 1) * half.min()
    end[600:800] = origin[600:800] + d * [1, 1]            # exact diagonals
    end[800:1000] = origin[800:1000] + d * [1, -1]
    origin, end = origin.astype(np.float32), end.astype(np.float32)
    g = api.Grid(sx, sy, res, rolling=1, max_range=1e9, raycast_impl=impl)
    g.raycast(origin, end)
    hits, misses = g.read_counts()
    eh, em, nupd = O.grid_raycast(og(g), origin, end)
    assert np.array_equal(hits, eh) and np.array_equal(misses, em)
    assert g.total_updates() == nupd
    g.close()


def test_raycast_scans_equals_explicit_rays():
    """slam_grid_raycast_scans_dev forms end = (float)(R p + t) itself."""
    batch = synth.make_batch(6, n_loop=256)
    Rt = [synth.pose_to_Rt(*p) for p in batch.true_poses]
    R = np.stack([r.reshape(4) for r, _ in Rt])
    t = np.stack([tt for _, tt in Rt])
    g = api.Grid(1000, 1000, 0.05, rolling=0)
    d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
    d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
    d_R = api.DeviceArray.from_host(R, np.float64)
    d_t = api.DeviceArray.from_host(t, np.float64)
    g.raycast_scans_dev(d_pts, d_off, batch.n_scans, batch.n_points, d_R, d_t)
    api.synchronize()
    hits, misses = g.read_counts()
    eh = np.zeros(g.cells, np.int32)
    em = np.zeros(g.cells, np.int32)
    total = 0
    for s in range(batch.n_scans):
        o, e = batch.scan_off[s], batch.scan_off[s + 1]
        end = O.transform_points(batch.pts[o:e], R[s], t[s])
        origin = np.tile(t[s].astype(np.float32), (e - o, 1))
        gp = og(g)
        _, _, n = O.grid_raycast(gp, origin, end, eh, em)
        total += n
    assert np.array_equal(hits, eh) and np.array_equal(misses, em) and g.total_updates() == total
    g.close()


def test_raycast_scans_ragged_batch():
    """Scans of very different sizes, empty ones among them: the point -> scan lookup of the pre-pass starts from a
    proportional guess and has to fall back to bisection here."""
    full = synth.make_batch(12, n_loop=256)
    sizes = [5, 1081, 0, 1, 700, 0, 0, 1081, 33, 2, 900, 64]
    pts, off = [], [0]
    for s, k in enumerate(sizes):
        o = full.scan_off[s]
        pts.append(full.pts[o:o + k])
        off.append(off[-1] + k)
    pts = np.concatenate(pts)
    off = np.array(off, np.int32)
    Rt = [synth.pose_to_Rt(*p) for p in full.true_poses]
    R = np.stack([r.reshape(4) for r, _ in Rt])
    t = np.stack([tt for _, tt in Rt])
    g = api.Grid(1000, 1000, 0.05, rolling=0)
    d_pts = api.DeviceArray.from_host(pts, np.float64)
    d_off = api.DeviceArray.from_host(off, np.int32)
    d_R = api.DeviceArray.from_host(R, np.float64)
    d_t = api.DeviceArray.from_host(t, np.float64)
    g.raycast_scans_dev(d_pts, d_off, len(sizes), len(pts), d_R, d_t)
    api.synchronize()
    hits, misses = g.read_counts()
    eh = np.zeros(g.cells, np.int32)
    em = np.zeros(g.cells, np.int32)
    total = 0
    for s in range(len(sizes)):
        o, e = off[s], off[s + 1]
        if e == o:
            continue
        end = O.transform_points(pts[o:e], R[s], t[s])
        origin = np.tile(t[s].astype(np.float32), (e - o, 1))
        _, _, n = O.grid_raycast(og(g), origin, end, eh, em)
        total += n
    assert np.array_equal(hits, eh) and np.array_equal(misses, em) and g.total_updates() == total
    g.close()


def test_finalize_matches_oracle_rule():
    rs = np.random.RandomState(11)
    g = api.Grid(80, 60, 0.5, min_cluster_points=3, rolling=1)
    obs = (rs.randn(3000, 3) * 4).astype(np.float32)
    gnd = (rs.randn(5000, 3) * 4).astype(np.float32)
    g.add_endpoints(obs, gnd)
    g.finalize()
    api.synchronize()
    hits, misses = g.read_counts()
    num, occ = O.grid_finalize(og(g), hits, misses)
    assert np.array_equal(g.read_occupancy(), occ)
    assert np.array_equal(g.read_num_pts(), num)
    assert set(np.unique(occ)) == {-1, 0, 100}
    g.close()


def test_inorder_mode_is_reference_order_exact():
    """mls.cpp:73-142 scan by scan: sequential += / -= on the double, thresholds."""
    rs = np.random.RandomState(12)
    g = api.Grid(200, 200, 0.2, min_cluster_points=20, rolling=1)   # local_mapper.cpp:29,86
    gp = og(g)
    num = np.zeros(g.cells)
    drv = np.full(g.cells, -1, np.int8)
    occ = np.full(g.cells, -1, np.int8)
    for scan in range(12):
        c = np.concatenate([rs.randn(2) * 3, [0, 0]])
        obs = (c + rs.randn(4000, 4) * [1.5, 1.5, 1, 1]).astype(np.float32)
        gnd = (rs.randn(6000, 4) * 6).astype(np.float32)
        g.add_scan_inorder(obs, gnd)
        O.grid_add_scan_inorder(gp, obs, gnd, num, drv, occ)
    assert np.array_equal(g.read_occupancy(), occ)
    assert np.array_equal(g.read_num_pts(), num)      # bit-exact doubles
    assert (occ == 100).sum() > 0 and (occ == 0).sum() > 0
    g.close()


def test_rolling_window_shift_and_clear():
    """MLS::setPose rolling, mls.cpp:408-479 + Grid::shiftOrigin mls.h:87-97."""
    rs = np.random.RandomState(13)
    sx = sy = 100
    res = 0.5
    g = api.Grid(sx, sy, res, rolling=1, min_cluster_points=2)
    gp = og(g)
    pts = (rs.rand(20000, 2) * 50 - 25).astype(np.float32)
    g.add_endpoints(pts, pts[:3000])
    h0, m0 = g.read_counts()
    g.set_pose(3.2, -1.8)                    # dx = round(6.4) = 6, dy = round(-3.6) = -4
    assert g.info()["origin_x"] == 6 and g.info()["origin_y"] == sy - 4
    px, py = g.get_pose()
    assert (px, py) == (6 * res, -4 * res)   # mls.cpp:430-431
    h1, m1 = g.read_counts()
    H0, H1 = h0.reshape(sy, sx), h1.reshape(sy, sx)
    # window cell (i,j) now shows what was at (i+6, j-4); cells rolled in are empty
    exp = np.zeros_like(H0)
    exp[4:, :sx - 6] = H0[:sy - 4, 6:]
    assert np.array_equal(H1, exp)
    # updates after the shift land relative to the new window
    g.add_endpoints(pts[:500], np.zeros((0, 2), np.float32))
    h2, _ = g.read_counts()
    eh, _, _, _ = O.grid_add_endpoints(gp, pts[:500], np.zeros((0, 2), np.float32))
    assert np.array_equal(h2, exp.ravel() + eh)
    g.close()


@pytest.mark.parametrize("size,res", [(2000, 0.05), (4000, 0.05), (6000, 0.02)])
def test_config2_grid_full_size_properties(size, res):
    """2000 x 2000 @ 0.05 m (config 2), 4000 x 4000 (config 4: 1024 tiles, segment offsets computed by the raycast
    workgroups) and 6000 x 6000 (2209 tiles: the three-kernel work list) with 32 registered scans: tiled ==
    global atomics, counts are linear in the beam set, and one hit per kept beam."""
    batch = synth.make_batch(32, n_loop=256)
    Rt = [synth.pose_to_Rt(*p) for p in batch.true_poses]
    R = np.stack([r.reshape(4) for r, _ in Rt])
    t = np.stack([tt for _, tt in Rt])
    d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
    d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
    d_R = api.DeviceArray.from_host(R, np.float64)
    d_t = api.DeviceArray.from_host(t, np.float64)
    out = []
    for impl in (api.RAYCAST_TILED, api.RAYCAST_GLOBAL, api.RAYCAST_TILED_MERGE):
        g = api.Grid(size, size, res, rolling=0, min_cluster_points=20, raycast_impl=impl)
        g.raycast_scans_dev(d_pts, d_off, batch.n_scans, batch.n_points, d_R, d_t)
        api.synchronize()
        out.append(g.read_counts() + (g.total_updates(),))
        if impl == api.RAYCAST_TILED:   # linearity: a second pass doubles every count
            g.raycast_scans_dev(d_pts, d_off, batch.n_scans, batch.n_points, d_R, d_t)
            api.synchronize()
            h2, m2 = g.read_counts()
            assert np.array_equal(h2, 2 * out[0][0]) and np.array_equal(m2, 2 * out[0][1])
        g.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[2][0]) and np.array_equal(out[0][1], out[2][1])
    assert out[0][2] == out[1][2] == out[2][2] == out[0][0].sum() + out[0][1].sum()
    assert out[0][0].sum() == batch.n_points          # every beam ends inside this grid


@pytest.mark.parametrize("seg,wg,cap", [(8, 1, 0), (16, 2, 0), (64, 2, 0), (511, 2, 0), (16, 2, 3), (8, 1, 1), (300, 1, 40)])
def test_raycast_launch_shapes_are_invisible(seg, wg, cap):
    """The tiled raycast's work distribution -- blocks a workgroup takes from a tile's list at a time (8 ... 511: with 511 a
    visit's chunk log holds two chunks, so a tile is written back and revisited after every second), workgroups per CU,
    workgroups in all (down to ONE, which then walks every tile by itself) -- changes who adds what, never the sums: counts
    equal the global-atomics form's on 96 scans in a 1500 x 1500 grid, twice over (the second pass starts from used cursors)."""
    batch = synth.make_batch(96, n_loop=256)
    Rt = [synth.pose_to_Rt(*p) for p in batch.true_poses]
    R = np.stack([r.reshape(4) for r, _ in Rt])
    t = np.stack([tt for _, tt in Rt])
    d = [api.DeviceArray.from_host(a, dt) for a, dt in ((batch.pts, np.float64), (batch.scan_off, np.int32), (R, np.float64), (t, np.float64))]
    ref = api.Grid(1500, 1500, 0.05, rolling=0, min_cluster_points=20, raycast_impl=api.RAYCAST_GLOBAL)
    ref.raycast_scans_dev(d[0], d[1], batch.n_scans, batch.n_points, d[2], d[3])
    api.synchronize()
    H, M = ref.read_counts()
    ref.close()
    g = api.Grid(1500, 1500, 0.05, rolling=0, min_cluster_points=20, raycast_seg_items=seg, raycast_wg_per_cu=wg, raycast_max_workgroups=cap)
    for rep in (1, 2):
        g.raycast_scans_dev(d[0], d[1], batch.n_scans, batch.n_points, d[2], d[3])
        api.synchronize()
        h, m = g.read_counts()
        assert np.array_equal(h, rep * H) and np.array_equal(m, rep * M), (seg, wg, cap, rep)
    st = g.raycast_stats()
    assert st["items"] > 0 and st["tile_write_backs"] >= 1
    assert g.total_updates() == 2 * (int(H.sum()) + int(M.sum()))
    g.close()


def test_reserve_then_batches_of_growing_size():
    """slam_grid_reserve: the raycast's scratch sized once for the largest batch -- batches of growing size enqueued back to
    back on one stream (each would otherwise free and re-allocate the scratch under the one before it, with a device-wide
    wait) leave the counts the global-atomics form leaves; a batch beyond the reservation still grows it (correct, slower)."""
    batch = synth.make_batch(64, n_loop=256)
    Rt = [synth.pose_to_Rt(*p) for p in batch.true_poses]
    R = np.stack([r.reshape(4) for r, _ in Rt])
    t = np.stack([tt for _, tt in Rt])
    d = [api.DeviceArray.from_host(a, dt) for a, dt in ((batch.pts, np.float64), (batch.scan_off, np.int32), (R, np.float64), (t, np.float64))]
    sizes = [5, 17, 33, 64]

    def run(g, stream=None):
        for n in sizes:
            g.raycast_scans_dev(d[0], d[1], n, int(batch.scan_off[n]), d[2], d[3], stream)
        api.synchronize()
        return g.read_counts()

    ref = api.Grid(1500, 1500, 0.05, rolling=0, min_cluster_points=20, raycast_impl=api.RAYCAST_GLOBAL)
    H, M = run(ref)
    ref.close()
    for reserve in (batch.n_points, int(batch.scan_off[17]), 0):      # all of it, less than the largest batch, none
        g = api.Grid(1500, 1500, 0.05, rolling=0, min_cluster_points=20)
        g.reserve(reserve)
        st = api.Stream()
        h, m = run(g, st)
        assert np.array_equal(h, H) and np.array_equal(m, M), reserve
        g.close()


def test_finalize_reset_inside_a_replayed_graph():
    """slam_grid_finalize_reset alternates between two device range buffers on the HOST, call by call; recorded into a hipGraph
    its kernel arguments would be frozen and every replay would read the same stale range (rows never folded or never reset,
    silently).  Captured, the call takes its two-step form: a replayed [endpoints, finalize_reset] must leave exactly what the
    same calls leave when issued one by one."""
    rs = np.random.RandomState(5)
    obs = (rs.randn(30000, 3) * [6, 4, 1]).astype(np.float32)
    gnd = (rs.randn(50000, 3) * [9, 9, 1]).astype(np.float32)
    d_obs, d_gnd = api.DeviceArray.from_host(obs), api.DeviceArray.from_host(gnd)
    L = api.lib()
    st = api.Stream()
    grids = [api.Grid(500, 500, 0.1, rolling=0, min_cluster_points=3) for _ in range(2)]

    def step(g):
        api.check(L.slam_grid_add_endpoints_dev(g.h, d_obs.ptr, len(obs), d_gnd.ptr, len(gnd), 3, st.ptr))
        g.finalize_reset(st)
    for g in grids:                       # the same uncaptured history on both
        step(g)
    st.synchronize()
    graph = api.Graph(st)
    with graph:
        step(grids[0])
    for r in range(3):
        graph.launch()
        step(grids[1])
        st.synchronize()
        h0, m0 = grids[0].read_counts()
        assert not h0.any() and not m0.any(), r                       # folded and reset by every replay
        assert np.array_equal(grids[0].read_occupancy(), grids[1].read_occupancy()), r
        assert np.array_equal(grids[0].read_num_pts(), grids[1].read_num_pts()), r
    step(grids[0])                        # ... and the handle goes on uncaptured afterwards
    step(grids[1])
    st.synchronize()
    assert np.array_equal(grids[0].read_occupancy(), grids[1].read_occupancy())
    assert (grids[0].read_occupancy() == 100).sum() > 0
    assert grids[0].total_updates() == grids[1].total_updates()
    for g in grids:
        g.close()


@pytest.mark.gpu
def test_event_query_does_not_block_and_tells_the_truth():
    """slam_event_query: hipEventQuery behind the C-ABI -- false while the work recorded before the event runs, true once it has
    finished, true for an event never recorded; what a caller that pipelines batches asks before it enqueues a wait packet."""
    import time
    g = api.Grid(2000, 2000, 0.05, rolling=0, min_cluster_points=20)
    st, ev, fresh = api.Stream(), api.Event(), api.Event()
    assert fresh.query() is True
    rs = np.random.RandomState(3)
    org = np.zeros((400000, 2), np.float32)
    end = (rs.uniform(-45, 45, (400000, 2))).astype(np.float32)
    d_org, d_end = api.DeviceArray.from_host(org, np.float32), api.DeviceArray.from_host(end, np.float32)
    seen_false = False
    for _ in range(20):                         # a few hundred microseconds of work per call: the query comes back while it runs
        g.raycast_dev(d_org, d_end, len(end), st)
    ev.record(st)
    t0 = time.perf_counter()
    while not ev.query():
        seen_false = True
        assert time.perf_counter() - t0 < 30.0
    st.synchronize()
    assert ev.query() is True and seen_false
    g.close()


@pytest.mark.gpu
def test_transform_cloud_is_the_host_loop_bit_for_bit():
    """slam_grid_transform_cloud_dev (MLS::addToMap's pcl::transformPointCloud, mls.cpp:34-53): (float)(r0 x + r1 y + r2 z + t) per
    coordinate in double, the products and sums rounded one by one as the adapter's host loop (and numpy) round them."""
    import ctypes as C
    rs = np.random.RandomState(4)
    for n, stride in ((1, 3), (1000, 3), (131072, 4), (777, 5)):
        cloud = (rs.randn(n, stride) * 30).astype(np.float32)
        cloud[::97, 0] = np.nan
        th = rs.uniform(-3, 3)
        q = np.array([0.01, -0.02, np.sin(th / 2), np.cos(th / 2)])
        q /= np.linalg.norm(q)
        x, y, z, w = q
        R = np.array([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], np.float64)
        t = rs.uniform(-0.2, 0.2, 3)
        d_in = api.DeviceArray.from_host(cloud)
        d_out = api.DeviceArray((n, 3), np.float32)
        api.check(api.lib().slam_grid_transform_cloud_dev(d_in.ptr, n, stride, R.ctypes.data_as(C.POINTER(C.c_double)),
                                                          t.ctypes.data_as(C.POINTER(C.c_double)), d_out.ptr, None))
        api.synchronize()
        p = cloud[:, :3].astype(np.float64)
        want = np.stack([((R[3 * k] * p[:, 0] + R[3 * k + 1] * p[:, 1]) + R[3 * k + 2] * p[:, 2]) + t[k] for k in range(3)], axis=1).astype(np.float32)
        assert np.array_equal(d_out.download(), want, equal_nan=True), (n, stride)


@pytest.mark.gpu
def test_host_is_pinned_tells_pinned_from_pageable():
    """slam_host_is_pinned: what the C++ adapters ask before an upload (a copy from pinned memory only enqueues)."""
    L = api.lib()
    pin = api.PinnedArray((1024,), np.float32)
    assert L.slam_host_is_pinned(pin.array.ctypes.data) == 1
    assert L.slam_host_is_pinned(pin.array.ctypes.data + 400) == 1          # anywhere inside the block
    page = np.zeros(1024, np.float32)
    assert L.slam_host_is_pinned(page.ctypes.data) == 0
    dev = api.DeviceArray((16,), np.float32)
    assert L.slam_host_is_pinned(dev.ptr) == 0                             # device memory is not host memory
    assert L.slam_host_is_pinned(None) == 0
