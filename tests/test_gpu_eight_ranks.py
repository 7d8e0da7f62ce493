"""BASELINE config 4 AS STATED -- 8 shards of 1024 scans, per-shard ICP + local grid, the shards' count planes summed into one
4000 x 4000 @ 0.05 m map -- run by LIBRARY code with eight ranks on the one GPU of the box: eight host threads of
one process, each with its own handle set (ICP model, grid, stream) and its own communicator over the library's host-staged
transport (slam_comm_create_host; the threads meet in a barrier where RCCL's ranks would meet in the all-reduce -- RCCL does not
put two ranks on one device, and a box allows six processes on its card).  What it pins: slam_grid_merge_begin / _finish with
EIGHT ranks at config 4's size -- the united row range, each rank's rows summed exactly once -- against the oracle's sequential
Bresenham of all 8 192 scans; and eight host threads driving the library at once."""
import threading

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth

pytestmark = pytest.mark.gpu

WORLD, S, GRID, RES, ITERS = 8, 1024, 4000, 0.05, 30


class ThreadRanks:
    """An all-reduce among the threads of one process: what the host transport's callback is given."""

    def __init__(self, world):
        self.world, self.bar, self.bufs, self.red, self.calls = world, threading.Barrier(world), [None] * world, None, 0

    def allreduce(self, rank):
        def f(a, op):
            self.bufs[rank] = a
            self.bar.wait(timeout=300)
            if rank == 0:
                stack = np.stack(self.bufs)
                self.red = np.add.reduce(stack, axis=0, dtype=np.int32) if op == api.COMM_SUM else stack.min(axis=0)
                self.calls += 1
            self.bar.wait(timeout=300)
            a[:] = self.red
            self.bar.wait(timeout=300)
        return f


def test_config4_as_stated_eight_ranks_merge_into_4000x4000():
    m_ga, m_nga = synth.make_map(10000)
    shards = [synth.make_batch(S, n_loop=S * WORLD, first=r * S) for r in range(WORLD)]
    ranks = ThreadRanks(WORLD)
    out, errors = [None] * WORLD, [None] * WORLD

    def rank_main(r):
        try:
            api.set_device(0)                       # (the HIP device is per host thread)
            b = shards[r]
            comm = api.Comm.host(r, WORLD, ranks.allreduce(r))
            assert comm.info() == (r, WORLD)
            icp = api.Icp(m_ga, m_nga, max_iter=ITERS, min_delta=-1.0)
            grid = api.Grid(GRID, GRID, RES, rolling=0, min_cluster_points=20)
            st = api.Stream()
            d_pts = api.DeviceArray.from_host(b.pts, np.float64)
            d_off = api.DeviceArray.from_host(b.scan_off, np.int32)
            d_nga = api.DeviceArray.from_host(b.scan_nga, np.int32)
            d_R0, d_t0 = api.DeviceArray.from_host(b.R, np.float64), api.DeviceArray.from_host(b.t, np.float64)
            d_R, d_t = api.DeviceArray(b.R.shape, np.float64), api.DeviceArray(b.t.shape, np.float64)
            d_res = api.DeviceArray((S,), api.RESULT_DTYPE)
            icp.fit_batch_from_dev(d_pts, d_off, d_nga, S, d_R0, d_t0, d_R, d_t, 5.0, d_res, None, st)
            grid.raycast_scans_dev(d_pts, d_off, S, b.n_points, d_R, d_t, st)
            st.synchronize()
            own = grid.total_updates()
            own_rows = grid.dirty_rows()
            comm.merge_begin(grid, st)
            rows = comm.merge_finish(grid, st)
            st.synchronize()
            hits, misses = grid.read_counts()
            grid.finalize(st)
            st.synchronize()
            res = d_res.download()
            out[r] = dict(R=d_R.download(), t=d_t.download(), iters=res["iters"].copy(), own=own, own_rows=own_rows, rows=rows,
                          sum=int(hits.astype(np.int64).sum() + misses.astype(np.int64).sum()),
                          hits=hits if r in (0, WORLD - 1) else None, misses=misses if r in (0, WORLD - 1) else None,
                          occ=grid.read_occupancy() if r in (0, WORLD - 1) else None, stats=comm.stats())
            comm.close()
        except BaseException as ex:   # a rank that fails must not leave seven waiting in the barrier
            errors[r] = ex
            ranks.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    assert first is None, repr(first)
    assert all(e is None for e in errors), errors

    # every rank registered its own shard (all 30 iterations, poses at the truth) ...
    for r, o in enumerate(out):
        assert (o["iters"] == ITERS).all()
        assert np.abs(o["t"] - shards[r].true_poses[:, :2]).max() < 0.05
    # ... the merge united the ranks' row ranges and every rank ends with every rank's updates, each exactly once
    lo, hi = min(o["own_rows"][0] for o in out), max(o["own_rows"][1] for o in out)
    assert all(tuple(o["rows"]) == (lo, hi) for o in out)
    total = sum(o["own"] for o in out)
    assert all(o["sum"] == total for o in out)
    assert all(o["stats"]["n_ranks"] == WORLD and o["stats"]["merges"] == 1 and o["stats"]["rows"] == hi - lo + 1 for o in out)
    a, z = out[0], out[-1]
    assert np.array_equal(a["hits"], z["hits"]) and np.array_equal(a["misses"], z["misses"]) and np.array_equal(a["occ"], z["occ"])

    # the oracle's sequential Bresenham of ALL 8 192 scans from the registered poses, and its finalize of those counts
    gp = O.grid_params(GRID, GRID, RES, rolling=0, min_cluster_points=20)
    H, M = np.zeros(GRID * GRID, np.int32), np.zeros(GRID * GRID, np.int32)
    for r, o in enumerate(out):
        b = shards[r]
        scan = np.repeat(np.arange(S), np.diff(b.scan_off))
        Rs, ts = o["R"].reshape(S, 2, 2)[scan], o["t"][scan]
        end = np.stack([Rs[:, 0, 0] * b.pts[:, 0] + Rs[:, 0, 1] * b.pts[:, 1] + ts[:, 0],
                        Rs[:, 1, 0] * b.pts[:, 0] + Rs[:, 1, 1] * b.pts[:, 1] + ts[:, 1]], 1).astype(np.float32)
        O.grid_raycast(gp, ts.astype(np.float32), end, H, M, n_threads=8)
    assert int(H.astype(np.int64).sum() + M.astype(np.int64).sum()) == total
    assert np.array_equal(a["hits"].reshape(-1), H) and np.array_equal(a["misses"].reshape(-1), M)
    num, occ = np.zeros(GRID * GRID), np.full(GRID * GRID, -1, np.int8)
    O.grid_finalize(gp, H, M, num, occ)
    assert np.array_equal(a["occ"].reshape(-1), occ)


def test_config5_as_stated_eight_ranks_stream_with_periodic_merge():
    """BASELINE config 5 across ranks: ONE loop of 10 240 scans, eight ranks streaming 1 280 each through slam_mapper_* (chunks of
    256 from pinned memory, sliding-window target rebuilt on the device every 2 chunks, dirty-row merge over the ranks every 2
    chunks and at the end), eight host threads on the one GPU over the host-staged communicator.  Every rank ends with the
    oracle's Bresenham of ALL 10 240 scans (from the registered poses), each update exactly once."""
    world, chunk, n_chunks, size = 8, 256, 5, 2000
    m_ga, m_nga = synth.make_map(5000)
    streams = [[synth.make_batch(chunk, n_loop=world * n_chunks * chunk, first=(r * n_chunks + k) * chunk) for k in range(n_chunks)]
               for r in range(world)]
    ranks = ThreadRanks(world)
    out, errors = [None] * world, [None] * world

    def rank_main(r):
        try:
            api.set_device(0)
            comm = api.Comm.host(r, world, ranks.allreduce(r))
            mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=size, grid_size_y=size, resolution=RES,
                            max_scans=chunk, max_points=max(c.n_points for c in streams[r]), icp=dict(max_iter=ITERS, min_delta=-1.0),
                            window_chunks=2, rebuild_every=2, keep_prior=1, target_points=5000, thin_res=0.1, merge_every=2)
            mp.use_comm(comm)
            R, t = np.zeros((n_chunks * chunk, 4)), np.zeros((n_chunks * chunk, 2))
            pending = []
            for k in range(n_chunks):
                if len(pending) == mp.n_slots:
                    slot, j = pending.pop(0)
                    R[j * chunk:(j + 1) * chunk], t[j * chunk:(j + 1) * chunk] = mp.wait(slot)
                pending.append((mp.push(streams[r][k]), k))
            for slot, j in pending:
                R[j * chunk:(j + 1) * chunk], t[j * chunk:(j + 1) * chunk] = mp.wait(slot)
            mp.finish()
            hits, misses = mp.grid.read_counts()
            st = mp.stats()
            out[r] = dict(R=R, t=t, stats=st, sum=int(hits.astype(np.int64).sum() + misses.astype(np.int64).sum()),
                          hits=hits if r in (0, world - 1) else None, misses=misses if r in (0, world - 1) else None,
                          occ=mp.grid.read_occupancy() if r in (0, world - 1) else None)
            mp.close()
            comm.close()
        except BaseException as ex:
            errors[r] = ex
            ranks.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t_ in threads:
        t_.start()
    for t_ in threads:
        t_.join(timeout=600)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    assert first is None, repr(first)
    assert all(e is None for e in errors), errors

    for r, o in enumerate(out):
        truth = np.concatenate([c.true_poses[:, :2] for c in streams[r]])
        assert np.abs(o["t"] - truth).max() < 0.05
        assert o["stats"]["chunks"] == n_chunks and o["stats"]["merges"] == n_chunks // 2 + 1 and o["stats"]["rebuilds"] >= 1
    assert all(o["sum"] == out[0]["sum"] for o in out)
    a, z = out[0], out[-1]
    assert np.array_equal(a["hits"], z["hits"]) and np.array_equal(a["misses"], z["misses"]) and np.array_equal(a["occ"], z["occ"])

    gp = O.grid_params(size, size, RES, rolling=0, min_cluster_points=20)
    H, M = np.zeros(size * size, np.int32), np.zeros(size * size, np.int32)
    for r, o in enumerate(out):
        for k, b in enumerate(streams[r]):
            scan = np.repeat(np.arange(chunk), np.diff(b.scan_off))
            Rs, ts = o["R"][k * chunk:(k + 1) * chunk].reshape(chunk, 2, 2)[scan], o["t"][k * chunk:(k + 1) * chunk][scan]
            end = np.stack([Rs[:, 0, 0] * b.pts[:, 0] + Rs[:, 0, 1] * b.pts[:, 1] + ts[:, 0],
                            Rs[:, 1, 0] * b.pts[:, 0] + Rs[:, 1, 1] * b.pts[:, 1] + ts[:, 1]], 1).astype(np.float32)
            O.grid_raycast(gp, ts.astype(np.float32), end, H, M, n_threads=8)
    assert int(H.astype(np.int64).sum() + M.astype(np.int64).sum()) == a["sum"]
    assert np.array_equal(a["hits"].reshape(-1), H) and np.array_equal(a["misses"].reshape(-1), M)
    num, occ = np.zeros(size * size), np.full(size * size, -1, np.int8)
    O.grid_finalize(gp, H, M, num, occ)
    assert np.array_equal(a["occ"].reshape(-1), occ)
