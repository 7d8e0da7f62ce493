import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth
m_ga, m_nga = synth.make_map()
batch = synth.make_batch(256)
# idle streams made before the mapper's: HIP deals streams over a fixed number of hardware queues (4 unless
# GPU_MAX_HW_QUEUES says otherwise), and two of the mapper's three streams on one queue cannot overlap
extra = [api.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), max_scans=256, max_points=batch.n_points, icp=dict(max_iter=30, min_delta=-1.0, pair_scans=int(sys.argv[3]) if len(sys.argv) > 3 else 0), slots=int(sys.argv[2]) if len(sys.argv) > 2 else 3)
for s in [mp.push(batch) for _ in range(mp.n_slots)]: mp.wait(s)
tp, tw = [], []
pending = []
t_all = time.perf_counter()
for k in range(20):
    if len(pending) == mp.n_slots:
        t0 = time.perf_counter(); mp.wait(pending.pop(0)); tw.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); pending.append(mp.push(batch)); tp.append(time.perf_counter() - t0)
for s in pending: mp.wait(s)
mp.finish()
print("%d idle streams, %d slots:" % (len(extra), mp.n_slots), "per chunk %.3f ms; push host time %.3f ms (median), wait %.3f ms" % ((time.perf_counter() - t_all) / 20 * 1e3, np.median(tp) * 1e3, np.median(tw) * 1e3))
