"""Round 5: bench.py --config 5 --chunk 512 took 2.3-2.5 ms per chunk with torch imported and 0.57 without.  Where the time is."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if "notorch" not in sys.argv:
    import torch
    torch.cuda.init()
import numpy as np
from slam_amd import api, synth
chunk = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
n_chunks = 10240 // chunk
api.set_device(0)
chunks = [synth.make_batch(chunk, n_loop=n_chunks * chunk, first=k * chunk) for k in range(n_chunks)]
m_ga, m_nga = synth.make_map(5000)
mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=2000, grid_size_y=2000, resolution=0.05, max_scans=chunk,
                max_points=max(c.n_points for c in chunks), icp=dict(max_iter=30, min_delta=-1.0), window_chunks=0, merge_every=8)
for s in [mp.push(chunks[0]) for _ in range(mp.n_slots)]:
    mp.wait(s)
api.synchronize()
t0 = time.perf_counter()
pending, marks = [], []
for k in range(n_chunks):
    if len(pending) == mp.n_slots:
        mp.wait(pending.pop(0))
    a = time.perf_counter()
    pending.append(mp.push(chunks[k]))
    marks.append((k, (a - t0) * 1e3, (time.perf_counter() - a) * 1e3))
t1 = time.perf_counter()
for s in pending:
    a = time.perf_counter(); mp.wait(s); marks.append(("drain", (a - t0) * 1e3, (time.perf_counter() - a) * 1e3))
t2 = time.perf_counter()
mp.finish()
t3 = time.perf_counter()
api.synchronize()
t4 = time.perf_counter()
print("chunk %d torch %s: push loop %.2f ms, drain %.2f, finish %.2f, sync %.2f; per chunk %.4f" % (chunk, "notorch" not in sys.argv, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) / n_chunks * 1e3))
print(" slowest calls:", sorted(marks, key=lambda m: -m[2])[:4])
mp.close()
