"""slam_grid_merge_async (include/slam_mi355x_rccl.h): the merge of SURVEY 8(e) -- united dirty rows, integer sum of those rows of
the [hits | misses] planes -- issued by the communicator's helper thread instead of the caller's, with what follows it on the
stream (finalize_reset / fold + finalize).  The result must be the synchronous merge's, bit for bit; the ranks' collectives must
meet in the order posted; a rank that never posts its merge must end the others with SLAM_E_TIMEOUT, not leave them waiting."""
import threading
import time

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth
from test_gpu_eight_ranks import ThreadRanks

pytestmark = pytest.mark.gpu

GRID, RES = 800, 0.125


def _dev_batch(b):
    R = np.stack([synth.pose_to_Rt(*p)[0].reshape(4) for p in b.true_poses])
    t = np.stack([synth.pose_to_Rt(*p)[1] for p in b.true_poses])
    return [api.DeviceArray.from_host(a, dt) for a, dt in ((b.pts, np.float64), (b.scan_off, np.int32), (R, np.float64), (t, np.float64))], R, t


def _oracle_planes(batches):
    gp = O.grid_params(GRID, GRID, RES, rolling=0, min_cluster_points=20)
    H, M = np.zeros(GRID * GRID, np.int32), np.zeros(GRID * GRID, np.int32)
    for b in batches:
        for s in range(b.n_scans):
            o, e = b.scan_off[s], b.scan_off[s + 1]
            R, t = synth.pose_to_Rt(*b.true_poses[s])
            O.grid_raycast(gp, np.tile(t.astype(np.float32), (e - o, 1)), O.transform_points(b.pts[o:e], R, t), H, M)
    return gp, H, M


def test_async_merge_over_rccl_equals_the_synchronous_one():
    """One rank over a real RCCL communicator: six steps alternating over two grids and two streams as bench.py's N > 1 pipeline
    posts them, once through merge_begin / merge_finish + finalize_reset on the caller's thread, once through merge_async."""
    comm = api.Comm(api.Comm.unique_id(), 0, 1)
    batches = [synth.make_batch(6, n_loop=96, first=6 * k) for k in range(6)]
    dev = [_dev_batch(b)[0] for b in batches]
    out = {}
    for how in ("sync", "async"):
        grids = [api.Grid(GRID, GRID, RES, rolling=0, min_cluster_points=3) for _ in range(2)]
        streams = [api.Stream(), api.Stream()]
        tickets, rows = {}, []
        for k, b in enumerate(batches):
            g, st, d = grids[k % 2], streams[k % 2], dev[k]
            if k % 2 in tickets:
                rows.append(comm.ticket_wait(tickets.pop(k % 2)))
            g.raycast_scans_dev(d[0], d[1], b.n_scans, b.n_points, d[2], d[3], st)
            if how == "sync":
                comm.merge_begin(g, st)
                rows.append(comm.merge_finish(g, st))
                g.finalize_reset(st)
            else:
                tickets[k % 2] = comm.merge_async(g, st, api.MERGE_THEN_FINALIZE_RESET)
        for i in sorted(tickets):
            rows.append(comm.ticket_wait(tickets[i]))
        comm.drain()
        api.synchronize()
        out[how] = ([g.read_occupancy() for g in grids], [g.read_num_pts() for g in grids], [g.read_counts() for g in grids], sorted(rows))
        for g in grids:
            g.close()
    for a, b in zip(out["sync"][0] + out["sync"][1], out["async"][0] + out["async"][1]):
        assert np.array_equal(a, b)
    assert all((h == 0).all() and (m == 0).all() for h, m in out["async"][2])      # finalize_reset ran behind every merge
    assert out["sync"][3] == out["async"][3] and all(0 <= lo <= hi < GRID for lo, hi in out["async"][3])
    st = comm.stats()
    assert st["async_merges"] == 6 and st["merges"] == 12
    comm.close()


def test_async_merge_three_thread_ranks_sum_every_update_once():
    """Three ranks (host threads, the host-staged transport) post two merges each through their helper threads: every rank ends
    with the oracle's planes of ALL ranks' scans, the done event and the fold + finalize continuation included."""
    world = 3
    ranks = ThreadRanks(world)
    shards = [[synth.make_batch(5, n_loop=120, first=(2 * r + j) * 5) for j in range(2)] for r in range(world)]
    out, errors = [None] * world, [None] * world

    def rank_main(r):
        try:
            api.set_device(0)
            comm = api.Comm.host(r, world, ranks.allreduce(r))
            comm.set_timeout(120.0)
            g = api.Grid(GRID, GRID, RES, rolling=0, min_cluster_points=20)
            g.enable_accumulator()
            st = api.Stream()
            done = api.Event()
            rows = []
            for b in shards[r]:
                d, _, _ = _dev_batch(b)
                g.raycast_scans_dev(d[0], d[1], b.n_scans, b.n_points, d[2], d[3], st)
                t = comm.merge_async(g, st, api.MERGE_THEN_FOLD_FINALIZE, done)
                rows.append(comm.ticket_wait(t))      # (the grid and its stream are the caller's again)
                done.synchronize()
            comm.drain()
            st.synchronize()
            out[r] = dict(occ=g.read_occupancy(), num=g.read_num_pts(), rows=rows, stats=comm.stats())
            comm.close()
            g.close()
        except BaseException as ex:
            errors[r] = ex
            ranks.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    assert first is None, repr(first)
    gp, H, M = _oracle_planes([b for sh in shards for b in sh])
    num, occ = np.zeros(GRID * GRID), np.full(GRID * GRID, -1, np.int8)
    O.grid_finalize(gp, H, M, num, occ)
    for o in out:
        assert np.array_equal(o["occ"].reshape(-1), occ)
        assert o["rows"] == out[0]["rows"] and o["stats"]["async_merges"] == 2 and o["stats"]["n_ranks"] == world


def test_a_rank_that_never_merges_times_the_others_out():
    """Fail fast (VERDICT r4 #1c): two thread-ranks, only rank 0 posts its merge.  Its helper thread's wait for the ranks' minimum
    is the host transport's here -- a barrier that breaks after 2 s -- and over RCCL the polled event with the communicator's
    time-out: either way the merge FAILS with SLAM_E_COMM / SLAM_E_TIMEOUT on the ticket, and every later call says the same."""
    ranks = ThreadRanks(2)

    def impatient(a, op):
        try:
            ranks.bar.wait(timeout=2.0)
        except threading.BrokenBarrierError:
            raise RuntimeError("the other rank never came")

    api.set_device(0)
    comm = api.Comm.host(0, 2, impatient)
    g = api.Grid(GRID, GRID, RES, rolling=0, min_cluster_points=20)
    st = api.Stream()
    b = synth.make_batch(3, n_loop=64)
    d, _, _ = _dev_batch(b)
    g.raycast_scans_dev(d[0], d[1], b.n_scans, b.n_points, d[2], d[3], st)
    t0 = time.monotonic()
    t = comm.merge_async(g, st, api.MERGE_THEN_NOTHING)
    with pytest.raises(api.SlamError) as ei:
        comm.ticket_wait(t)
    assert ei.value.code in (api.E_COMM, api.E_TIMEOUT) and time.monotonic() - t0 < 20.0
    with pytest.raises(api.SlamError) as ei2:
        comm.merge_async(g, st, api.MERGE_THEN_NOTHING)
    assert ei2.value.code == ei.value.code
    comm.close()
    g.close()


def test_range_timeout_over_rccl_event_poll():
    """The polled wait itself: a merge whose united range is held back on the device (its stream waits for an event nobody
    records until later) past the communicator's time-out fails with SLAM_E_TIMEOUT and names the rank."""
    comm = api.Comm(api.Comm.unique_id(), 0, 1)
    comm.set_timeout(1.0)
    g = api.Grid(GRID, GRID, RES, rolling=0, min_cluster_points=20)
    st, blocker = api.Stream(), api.Stream()
    spin = api.DeviceArray((1 << 24,), np.int32)
    gate = api.Event()
    # keep `st` busy for > 1 s: a long chain of memsets on another stream, `st` waits for its end
    for _ in range(4000):
        spin.zero(blocker)
    gate.record(blocker)
    st.wait_event(gate)
    t = comm.merge_async(g, st, api.MERGE_THEN_NOTHING)
    try:
        comm.ticket_wait(t)
        finished_in_time = True           # (a box fast enough to clear 256 GB of memsets in a second: nothing to assert)
    except api.SlamError as ex:
        finished_in_time = False
        assert ex.code == api.E_TIMEOUT and "rank 0" in str(ex)
    api.synchronize()
    if not finished_in_time:
        with pytest.raises(api.SlamError):
            comm.merge_begin(g, st)
    comm.close()
    g.close()
