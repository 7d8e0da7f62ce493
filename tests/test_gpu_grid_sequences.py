"""slam_grid_reset_counts, slam_grid_finalize and slam_grid_finalize_reset work on the rows the grid knows to be touched or changed, not on the
whole planes.  Random sequences of every operation that writes counts (raycast, endpoints, writes through the raw
plane pointer followed by slam_grid_mark_rows -- what a merge over the GPUs does), clears them (reset, fold into the
accumulator, clear), moves the window (set_pose on a rolling grid) or changes the rule (min_cluster_points) against
a shadow on the host: after EVERY finalize the whole evidence and occupancy planes equal the oracle's finalize of the
shadow's counts, and the counts themselves equal the shadow's."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api

from oracle_lib import roll

pytestmark = pytest.mark.gpu


class Shadow:
    def __init__(self, size, res, rolling, minp):
        self.size, self.res, self.rolling, self.minp = size, res, rolling, minp
        self.H = np.zeros((size, size), np.int32)
        self.M = np.zeros((size, size), np.int32)
        self.AH, self.AM = np.zeros_like(self.H), np.zeros_like(self.M)
        self.cx = self.cy = 0.0

    def params(self):
        return O.grid_params(self.size, self.size, self.res, max_range=0.45 * self.size * self.res, rolling=self.rolling,
                             min_cluster_points=self.minp, pose_x=self.cx, pose_y=self.cy)

    def expected(self):
        num, occ = np.zeros(self.size ** 2), np.full(self.size ** 2, -1, np.int8)
        O.grid_finalize(self.params(), (self.H + self.AH).reshape(-1), (self.M + self.AM).reshape(-1), num, occ)
        return num, occ


def check(g, sh, what):
    g.finalize()
    api.synchronize()
    num, occ = sh.expected()
    hits, misses = g.read_counts()
    assert np.array_equal(hits, (sh.H + sh.AH).reshape(-1)) and np.array_equal(misses, (sh.M + sh.AM).reshape(-1)), what
    assert np.array_equal(g.read_occupancy(), occ), what
    assert np.array_equal(g.read_num_pts(), num), what


@pytest.mark.parametrize("rolling,seed", [(r, s) for r in (0, 1) for s in range(1, 13)])
def test_random_sequences_against_a_host_shadow(rolling, seed):
    rs = np.random.RandomState(seed)
    size, res = 300, 0.1
    sh = Shadow(size, res, rolling, 3)
    g = api.Grid(size, size, res, rolling=rolling, min_cluster_points=3, max_range=0.45 * size * res)
    g.enable_accumulator()
    ptr, n_ints = g.counts_dev()
    L = api.lib()
    log = []
    for step in range(120):
        op = rs.choice(["raycast", "raycast", "endpoints", "reset", "fold", "finalize", "finalize", "external", "minp", "clear", "roll",
                        "finalize_reset", "finalize_reset"])
        if op == "roll" and not rolling:
            op = "raycast"
        if op == "external" and rolling:      # (the raw pointer is in storage order: the shadow would need the torus too)
            op = "endpoints"
        log.append(op)
        if op == "raycast":       # a fan of beams from a random origin: a band of rows
            n = int(rs.randint(1, 200))
            o = np.tile(rs.uniform(-8, 8, 2) + [sh.cx, sh.cy], (n, 1)).astype(np.float32)
            e = (o + rs.uniform(-6, 6, (n, 2))).astype(np.float32)
            g.raycast(o, e)
            # (the oracle takes the same points as the library; on a rolling grid both subtract the window's pose)
            O.grid_raycast(sh.params(), o, e, sh.H.reshape(-1), sh.M.reshape(-1))
        elif op == "endpoints":
            obs = (rs.uniform(-10, 10, (int(rs.randint(0, 300)), 2)) + [sh.cx, sh.cy]).astype(np.float32)
            gnd = (rs.uniform(-10, 10, (int(rs.randint(0, 300)), 2)) + [sh.cx, sh.cy]).astype(np.float32)
            g.add_endpoints(obs, gnd)
            O.grid_add_endpoints(sh.params(), obs, gnd, sh.H.reshape(-1), sh.M.reshape(-1))
        elif op == "reset":
            g.reset_counts()
            sh.H[:] = 0
            sh.M[:] = 0
        elif op == "finalize_reset":   # one launch: evidence and occupancy of the counts so far, then the count planes zero
            num, occ = sh.expected()
            g.finalize_reset()
            api.synchronize()
            assert np.array_equal(g.read_occupancy(), occ) and np.array_equal(g.read_num_pts(), num), "finalize_reset after %s" % " ".join(log[-12:])
            sh.H[:] = 0
            sh.M[:] = 0
            hits, misses = g.read_counts()
            assert np.array_equal(hits, sh.AH.reshape(-1)) and np.array_equal(misses, sh.AM.reshape(-1))
        elif op == "fold":        # the rows that hold counts (what a merge returns) -- or, sometimes, only some of them
            rows = np.flatnonzero((sh.H != 0).any(1) | (sh.M != 0).any(1))
            if len(rows) == 0:
                g.fold(0, -1)
                continue
            lo, hi = int(rows.min()), int(rows.max())
            if rs.rand() < 0.3 and hi > lo:
                hi = int(rs.randint(lo, hi))
            oy = g.info()["origin_y"]          # slam_grid_fold takes STORAGE rows (what the dirty range and a merge deliver)
            slo, shi = (lo + oy) % size, (hi + oy) % size
            if shi < slo:                      # the band wraps around the torus: two folds
                g.fold(slo, size - 1)
                g.fold(0, shi)
            else:
                g.fold(slo, shi)
            sh.AH[lo:hi + 1] += sh.H[lo:hi + 1]
            sh.AM[lo:hi + 1] += sh.M[lo:hi + 1]
            sh.H[lo:hi + 1] = 0
            sh.M[lo:hi + 1] = 0
        elif op == "external":    # another rank's counts summed into a band of rows behind the library's back, then marked
            lo = int(rs.randint(0, size - 1))
            hi = int(min(size - 1, lo + rs.randint(0, 40)))
            add = (rs.rand(hi - lo + 1, size) < 0.02).astype(np.int32) * rs.randint(1, 5)
            api.synchronize()
            for plane, arr in ((0, sh.H), (1, sh.M)):
                arr[lo:hi + 1] += add
                band = np.ascontiguousarray(arr[lo:hi + 1])
                api.check(L.slam_memcpy_h2d(C.c_void_p(ptr + 4 * (plane * size * size + lo * size)), band.ctypes.data, band.nbytes, None))
            g.mark_rows(lo, hi)
        elif op == "minp":
            sh.minp = int(rs.randint(1, 6))
            g.set_min_cluster_points(sh.minp)
        elif op == "clear":
            g.clear()
            sh.H[:] = 0
            sh.M[:] = 0
            sh.AH[:] = 0
            sh.AM[:] = 0
        elif op == "roll":
            nx, ny = sh.cx + rs.randint(-15, 16) * res, sh.cy + rs.randint(-15, 16) * res
            g.set_pose(nx, ny)
            dx, dy = int(np.round((nx - sh.cx) / res)), int(np.round((ny - sh.cy) / res))
            for name in ("H", "M", "AH", "AM"):
                setattr(sh, name, roll(getattr(sh, name), dx, dy))
            sh.cx += dx * res
            sh.cy += dy * res
        if op == "finalize" or step % 7 == 6:
            check(g, sh, "after %s" % " ".join(log[-12:]))
    check(g, sh, "at the end")
    g.close()
