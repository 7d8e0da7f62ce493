"""Host-side timeline of the streaming mapper with a sliding target: per push / wait times, background rebuild on/off.
python tools/config5_trace.py [background=1] [chunks=40]"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth

kw = dict(a.split("=") for a in sys.argv[1:])
bg, n_chunks, chunk = int(kw.get("background", 1)), int(kw.get("chunks", 40)), 256
slots, pair = int(kw.get("slots", 0)), int(kw.get("pair", 0))
chunks = [synth.make_batch(chunk, n_loop=n_chunks * chunk, first=k * chunk) for k in range(n_chunks)]
m_ga, m_nga = synth.make_map(5000)
mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), max_scans=chunk, max_points=max(c.n_points for c in chunks),
                icp=dict(max_iter=30, min_delta=-1.0, pair_scans=pair), slots=slots, window_chunks=4, rebuild_every=4, keep_prior=1, target_points=5000,
                thin_res=0.1, merge_every=8, background_rebuild=bg)
for s in [mp.push(chunks[0]) for _ in range(mp.n_slots)]:
    mp.wait(s)
api.synchronize()
tp, tw, pending = [], [], []
t_all = time.perf_counter()
for k in range(n_chunks):
    if len(pending) == mp.n_slots:
        t0 = time.perf_counter(); mp.wait(pending.pop(0)); tw.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); pending.append(mp.push(chunks[k])); tp.append(time.perf_counter() - t0)
for s in pending:
    mp.wait(s)
mp.finish()
dt = (time.perf_counter() - t_all) / n_chunks
print("background=%d slots=%d pair=%d: %.3f ms per chunk; stats %s" % (bg, mp.n_slots, pair, dt * 1e3, mp.stats()))
print(" push ms:", " ".join("%.2f" % (x * 1e3) for x in tp))
print(" wait ms:", " ".join("%.2f" % (x * 1e3) for x in tw))
