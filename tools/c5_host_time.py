#!/usr/bin/env python3
"""Config 5's producer loop with the host's time split: filling the slot's pinned buffers (numpy copies), slam_mapper_push, slam_mapper_wait."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import api, synth
import ctypes as C

kw = dict(a.split("=") for a in sys.argv[1:])
chunk, n_chunks = int(kw.get("chunk", 256)), int(kw.get("chunks", 40))
nofill = int(kw.get("nofill", 0))      # 1: every slot's pinned buffers are filled once (the device's rate without the producer's copies)
api.set_device(0)
chunks = [synth.make_batch(chunk, n_loop=n_chunks * chunk, first=k * chunk) for k in range(n_chunks)]
if int(kw.get("same_chunk", 0)):
    chunks = [synth.make_batch(chunk)] * n_chunks          # bench.py's config-2 batch, again and again (its stream_rate leg)
m_ga, m_nga = synth.make_map(int(kw.get("map_points", 5000)))
for window in [int(w) for w in kw.get("windows", "4,0").split(",")]:
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20), grid_size_x=2000, grid_size_y=2000, resolution=0.05, max_scans=chunk,
                    max_points=max(c.n_points for c in chunks), icp=dict(max_iter=30, min_delta=-1.0, list_min_halo=float(kw.get("min_halo", 0))), window_chunks=window, rebuild_every=int(kw.get("rebuild_every", 4)),
                    keep_prior=1, target_points=5000, thin_res=0.1, merge_every=int(kw.get("merge_every", 8)), registration_streams=int(kw.get("reg_streams", 0)),
                    slots=int(kw.get("slots", 0)))
    for _ in range(3):
        for s in [mp.push(chunks[0]) for _ in range(mp.n_slots)]:
            mp.wait(s)
    api.synchronize()
    t_fill = t_push = t_wait = 0.0
    rows = []
    L = api.lib()
    t0 = time.perf_counter()
    pending = []
    for k in range(n_chunks):
        w_ = 0.0
        if len(pending) == mp.n_slots:
            a = time.perf_counter()
            mp.wait(pending.pop(0))
            w_ = time.perf_counter() - a
            t_wait += w_
        b = chunks[k]
        a = time.perf_counter()
        slot = C.c_int()
        api.check(L.slam_mapper_next_slot(mp.h, C.byref(slot)))
        pts, off, nga, R, t = mp._slot_views(slot.value)
        S, P = b.n_scans, b.n_points
        if nofill and k >= mp.n_slots:
            b = chunks[k % mp.n_slots]; S, P = b.n_scans, b.n_points
        else:
          pts[:2 * P] = b.pts.reshape(-1); off[:S + 1] = b.scan_off; nga[:S] = b.scan_nga
        R[:4 * S] = b.R.reshape(-1); t[:2 * S] = b.t.reshape(-1)
        c = time.perf_counter()
        out = C.c_int()
        api.check(L.slam_mapper_push(mp.h, S, P, 0.0, 0.0, C.byref(out)))
        mp._n = getattr(mp, "_n", {}); mp._n[out.value] = S
        d = time.perf_counter()
        t_fill += c - a; t_push += d - c
        rows.append((k, (a - t0) * 1e3, w_ * 1e3, (c - a) * 1e3, (d - c) * 1e3))
        pending.append(out.value)
    for s in pending:
        mp.wait(s)
    mp.finish()
    api.synchronize()
    el = time.perf_counter() - t0
    print(kw, mp.stats(), mp.target_index_info())
    print("window %d: %.4f ms per chunk; host per chunk: fill %.4f  push %.4f  wait %.4f  (slots %d)" %
          (window, el / n_chunks * 1e3, t_fill / n_chunks * 1e3, t_push / n_chunks * 1e3, t_wait / n_chunks * 1e3, mp.n_slots))
    for r in rows[8:28]:
        print("   chunk %2d at %7.3f ms: waited %.3f, fill %.3f, push %.3f" % r)
    mp.close()
