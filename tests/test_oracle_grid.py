"""Oracle checks for the occupancy-grid half (mls.cpp:59-150 restated; the
Bresenham traversal is the build's own definition)."""
import os

import numpy as np

import oracle_lib as O
from slam_amd import synth


def test_cell_index_truncates_toward_zero():
    # mls.cpp:77-78: (int)(pt.x/resolution + size_x/2)
    g = O.grid_params(10, 10, 1.0, rolling=1)
    assert O.grid_cell(g, 0.0, 0.0)[1:] == (5, 5)
    assert O.grid_cell(g, -4.5, 0.0)[1:] == (0, 5)     # 0.5 -> 0
    assert O.grid_cell(g, -5.5, 0.0)[0] == 0 + 10 * 5   # -0.5 -> 0, NOT floor(-0.5) = -1
    assert O.grid_cell(g, -6.0, 0.0)[0] == -1           # -1.0 -> -1: out
    assert O.grid_cell(g, 4.99, 4.99)[1:] == (9, 9)
    assert O.grid_cell(g, 5.0, 0.0)[0] == -1


def test_cell_index_odd_size_uses_integer_half():
    g = O.grid_params(11, 11, 0.5, rolling=1)            # size_x/2 == 5
    assert O.grid_cell(g, 0.0, 0.0)[1:] == (5, 5)
    assert O.grid_cell(g, 2.9, -2.4)[1:] == (10, 0)


def test_range_gate_rolling_uses_float_sqrt():
    # mls.cpp:82,90: sqrt(pt.x*pt.x + pt.y*pt.y) > max_range(75)
    g = O.grid_params(4000, 4000, 0.05, rolling=1)
    assert O.grid_cell(g, 75.0, 0.0)[0] >= 0             # == 75 is kept
    x = np.nextafter(np.float32(75.0), np.float32(100))
    # x*x rounds to a float whose float sqrt is exactly 75 -> kept; a double sqrt would reject
    s = np.float32(x * x)
    if np.sqrt(s, dtype=np.float32) == np.float32(75.0):
        assert O.grid_cell(g, float(x), 0.0)[0] >= 0
    assert O.grid_cell(g, 75.01, 0.0)[0] == -1
    assert O.grid_cell(g, 53.1, 53.1)[0] == -1


def test_range_gate_global_frame_uses_pose():
    # mls.cpp:84-86
    g = O.grid_params(4000, 4000, 0.05, rolling=0, pose_x=50.0, pose_y=0.0)
    assert O.grid_cell(g, -20.0, 0.0)[0] >= 0            # 70 m from the pose
    assert O.grid_cell(g, -26.0, 0.0)[0] == -1           # 76 m


def test_y_is_tested_against_size_x():
    # mls.cpp:90 "y >= size_x" (sic)
    g = O.grid_params(10, 20, 1.0, rolling=1)
    assert O.grid_cell(g, 0.0, 4.5)[0] == -1             # y = 14 >= size_x = 10
    assert O.grid_cell(g, 0.0, -0.5)[1:] == (5, 9)


def test_nan_and_huge_are_skipped():
    g = O.grid_params(10, 10, 1.0, rolling=1)
    assert O.grid_cell(g, float("nan"), 0.0)[0] == -1
    assert O.grid_cell(g, 1e30, 0.0)[0] == -1


def test_endpoint_counts_and_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "grid_golden.npz"))
    batch = synth.make_batch(1, n_loop=256)
    R, t = synth.pose_to_Rt(*G["true_pose"])
    end = O.transform_points(batch.pts, R, t)
    g = O.grid_params(500, 500, 0.1, min_cluster_points=20)
    hits, misses, cells, n = O.grid_add_endpoints(g, end[:600], end[600:])
    assert np.array_equal(cells, G["end_cells"]) and n == int(G["end_n"])
    assert hits.sum() == (cells[:600] >= 0).sum() and misses.sum() == (cells[600:] >= 0).sum()


def test_raycast_golden_and_invariants(golden_dir):
    G = np.load(os.path.join(golden_dir, "grid_golden.npz"))
    batch = synth.make_batch(1, n_loop=256)
    R, t = synth.pose_to_Rt(*G["true_pose"])
    end = O.transform_points(batch.pts, R, t)
    origin = np.tile(np.array(t, dtype=np.float32), (len(end), 1))
    g = O.grid_params(500, 500, 0.1, min_cluster_points=20)
    hits, misses, n = O.grid_raycast(g, origin, end)
    assert n == int(G["ray_n_upd"]) == hits.sum() + misses.sum()
    assert np.array_equal(np.flatnonzero(hits), G["ray_hit_cells"])
    assert np.array_equal(hits[G["ray_hit_cells"]], G["ray_hit_counts"])
    assert np.array_equal(np.flatnonzero(misses), G["ray_miss_cells"])
    assert np.array_equal(misses[G["ray_miss_cells"]], G["ray_miss_counts"])
    # one hit per beam, at the endpoint cell G1's formula gives
    eh, _, cells, _ = O.grid_add_endpoints(g, end, np.zeros((0, 2), np.float32))
    assert np.array_equal(eh, hits)


def test_bresenham_single_rays():
    g = O.grid_params(64, 64, 1.0, max_range=1e9, rolling=1)

    def walk(x0, y0, x1, y1):
        o = np.array([[x0 - 32 + 0.5, y0 - 32 + 0.5]], np.float32)
        e = np.array([[x1 - 32 + 0.5, y1 - 32 + 0.5]], np.float32)
        hits, misses, n = O.grid_raycast(g, o, e)
        return hits.reshape(64, 64), misses.reshape(64, 64), n

    for (x0, y0, x1, y1) in [(5, 5, 20, 9), (20, 9, 5, 5), (5, 5, 9, 20), (30, 40, 30, 10),
                             (7, 7, 7, 7), (0, 0, 63, 63), (63, 0, 0, 63), (10, 10, 11, 10)]:
        h, m, n = walk(x0, y0, x1, y1)
        L = max(abs(x1 - x0), abs(y1 - y0))
        assert n == L + 1 and h.sum() == 1 and h[y1, x1] == 1
        assert m.sum() == L and m.max() <= 1 and m[y1, x1] == 0
        if L:
            assert m[y0, x0] == 1
        ys, xs = np.nonzero(m)
        if abs(x1 - x0) >= abs(y1 - y0) and L:  # x-major: one cell per column
            assert len(set(xs)) == L
            # y_i = y0 + sy*floor((2 i dy + dx) / (2 dx))
            dx, dy = abs(x1 - x0), abs(y1 - y0)
            sx, sy = (1 if x1 > x0 else -1), (1 if y1 > y0 else -1)
            for i in range(L):
                assert m[y0 + sy * ((2 * i * dy + dx) // (2 * dx)), x0 + sx * i] == 1


def test_finalize_equals_inorder_for_one_scan():
    # within a scan all obstacle points precede all ground points
    rs = np.random.RandomState(11)
    g = O.grid_params(40, 40, 0.5, min_cluster_points=3, rolling=1)
    obs = (rs.randn(600, 3) * 2).astype(np.float32)
    gnd = (rs.randn(900, 3) * 2).astype(np.float32)
    hits, misses, _, _ = O.grid_add_endpoints(g, obs, gnd)
    num, occ = O.grid_finalize(g, hits, misses)
    num2 = np.zeros(1600)
    drv = np.full(1600, -1, np.int8)
    occ2 = np.full(1600, -1, np.int8)
    O.grid_add_scan_inorder(g, obs, gnd, num2, drv, occ2)
    assert np.array_equal(occ, occ2)
    assert np.abs(num - num2).max() < 1e-9  # last-ulp differences only
    assert set(np.unique(occ)) <= {-1, 0, 100}
