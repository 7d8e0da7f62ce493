#!/usr/bin/env python3
"""Config 5 without a profiler: the mapper's own device marks (HIP timing events, measurement build) beside its host marks.
    SLAM_AMD_MEASURE=1 SLAM_MAPPER_TRACE=gpurun_out/c5_trace.txt python tools/c5_device_timeline.py [window=4] [rebuild_every=4]
prints the registrations' start / duration / gap and what the other streams did around every rebuild."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLAM_AMD_MEASURE", "1")
os.environ.setdefault("SLAM_MAPPER_TRACE", os.path.join(ROOT, "gpurun_out", "c5_trace.txt"))
import numpy as np
from slam_amd import api, synth

kw = dict(a.split("=") for a in sys.argv[1:])
chunk, n_chunks = 256, int(kw.get("chunks", 24))
window, every = int(kw.get("window", 4)), int(kw.get("rebuild_every", 4))
api.set_device(0)
chunks = [synth.make_batch(chunk, n_loop=n_chunks * chunk, first=k * chunk) for k in range(n_chunks)]
m_ga, m_nga = synth.make_map(5000)
mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=2000, grid_size_y=2000, resolution=0.05, max_scans=chunk,
                max_points=max(c.n_points for c in chunks), icp=dict(max_iter=30, min_delta=-1.0, list_min_halo=float(kw.get("min_halo", 0))), window_chunks=window, rebuild_every=every,
                keep_prior=1, target_points=5000, thin_res=0.1, merge_every=8, registration_streams=int(kw.get("reg_streams", 0)),
                slots=int(kw.get("slots", 0)))
t0 = time.perf_counter()
pending = []
for k in range(n_chunks):
    if len(pending) == mp.n_slots:
        mp.wait(pending.pop(0))
    pending.append(mp.push(chunks[k]))
for s in pending:
    mp.wait(s)
mp.finish()
print("%.4f ms per chunk over %d chunks (the first rebuilds included)" % ((time.perf_counter() - t0) / n_chunks * 1e3, n_chunks), flush=True)
mp.close()
dev = [l.split() for l in open(os.environ["SLAM_MAPPER_TRACE"]) if l.startswith("DEV")]
ev = [(l[1], float(l[2])) for l in dev]
fits = []
cur = None
for name, t in ev:
    if name == "fit>":
        cur = t
    elif name == "fit<" and cur is not None:
        fits.append((cur, t))
        cur = None
print("registrations: start, duration, gap to the next (us)")
for (a, b), (c, d) in zip(fits, fits[1:]):
    print("  %9.1f %7.1f %7.1f" % (a, b - a, c - b))
# everything between consecutive marks of each kind
for kind in ("rebuild", "raycast", "copy"):
    st = [t for n, t in ev if n == kind + ">"]
    en = [t for n, t in ev if n == kind + "<"]
    print(kind, " ".join("%.0f-%.0f" % (a, b) for a, b in zip(st, en)))
