"""Randomised streams through slam_mapper_* with a sliding-window target: window length, rebuild cadence, slots, one or two
registration streams, thinning or stride, strict or lagging adoption, chunk size (the spread form for small chunks) -- every
run must register every scan within a few centimetres of the truth (a window entry overwritten under a rebuild that still
reads it, or a target adopted half-built, shows as scans that drift off), and the strict runs must repeat bit for bit.
A soak to run by hand after touching the mapper.    timeout -k 10 600 python tools/soak_mapper.py [seconds]"""
import sys
import time

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests'))
import numpy as np
from slam_amd import api, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(23)
m_ga, m_nga = synth.make_map(10000)
t_end = time.time() + budget
n = 0


def run(batch, chunk, kw):
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=1200, grid_size_y=1200, resolution=0.1,
                    max_scans=chunk, max_points=chunk * 1100, icp=dict(max_iter=20, min_delta=1e-6), **kw)
    R, t = np.zeros((batch.n_scans, 4)), np.zeros((batch.n_scans, 2))
    pending = []
    for s0 in range(0, batch.n_scans, chunk):
        s1 = min(s0 + chunk, batch.n_scans)
        if len(pending) == mp.n_slots:
            slot, a, b = pending.pop(0)
            R[a:b], t[a:b] = mp.wait(slot)
        o, e = batch.scan_off[s0], batch.scan_off[s1]
        c = synth.ScanBatch(batch.pts[o:e], (batch.scan_off[s0:s1 + 1] - o).astype(np.int32), batch.scan_nga[s0:s1],
                            batch.R[s0:s1], batch.t[s0:s1], batch.true_poses[s0:s1])
        pending.append((mp.push(c), s0, s1))
    for slot, a, b in pending:
        R[a:b], t[a:b] = mp.wait(slot)
    mp.finish()
    st = mp.stats()
    H, M = mp.grid.read_counts()
    mp.close()
    return R, t, st, H, M


while time.time() < t_end:
    chunk = int(rs.choice([4, 12, 24, 64, 128]))
    n_scans = chunk * int(rs.randint(6, 20))
    batch = synth.make_batch(n_scans, n_loop=max(256, n_scans), first=int(rs.randint(0, 100)))
    strict = int(rs.rand() < 0.35)
    kw = dict(window_chunks=int(rs.randint(1, 6)), rebuild_every=int(rs.randint(1, 6)), keep_prior=1, target_points=int(rs.choice([1500, 6000, 12000])),
              thin_res=float(rs.choice([0.0, 0.1, 0.25])), merge_every=int(rs.choice([0, 3, 8])), strict_window=strict,
              background_rebuild=int(rs.rand() < 0.8), slots=int(rs.choice([0, 2, 3, 5, 8])), registration_streams=int(rs.choice([0, 1, 2])))
    R, t, st, H, M = run(batch, chunk, kw)
    err = np.abs(t - batch.true_poses[:, :2]).max()
    assert err < 0.08, ("pose error", err, chunk, n_scans, kw, st)
    assert H.sum() > 0.5 * batch.n_points, ("hits", int(H.sum()), batch.n_points, kw)
    if strict:
        R2, t2, _, H2, M2 = run(batch, chunk, kw)
        assert np.array_equal(R, R2) and np.array_equal(t, t2) and np.array_equal(H, H2) and np.array_equal(M, M2), ("strict runs differ", chunk, n_scans, kw)
    n += 1
print("soak ok: %d streams in %.0f s" % (n, budget))
