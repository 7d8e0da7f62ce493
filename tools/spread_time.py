#!/usr/bin/env python3
"""The spread form (slam_amd/csrc/icp_single.hip) one scan at a time, kernel time by HIP events: config 3's match
(tests/golden/spread_case3.npz: a 20 870-point lidar target in HBM/L2, voxel-filtered scenes of ~600 points, reference defaults
max_iter 20 / min_delta 1e-6) and a 1081-beam scan against the 2 x 19 999 room model (the CCICP cap) and the 10 k map.
Every pose is held against the CPU oracle.  With SLAM_AMD_MEASURE=1 SLAM_SPREAD_STAMPS=1 it also prints where an
iteration's time goes (in-kernel wall-clock stamps).  `python tools/spread_time.py [reps]`"""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from slam_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def time_fit(icp, s_ga, s_nga, R0, t0, reps, max_iter=20):
    pts = np.concatenate([np.asarray(s_ga, np.float64).reshape(-1, 2), np.asarray(s_nga, np.float64).reshape(-1, 2)])
    d_pts = api.DeviceArray.from_host(pts, np.float64)
    d_off = api.DeviceArray.from_host(np.array([0, len(pts)], np.int32))
    d_nga = api.DeviceArray.from_host(np.array([len(s_ga)], np.int32))
    d_R0 = api.DeviceArray.from_host(np.asarray(R0, np.float64).reshape(1, 4))
    d_t0 = api.DeviceArray.from_host(np.asarray(t0, np.float64).reshape(1, 2))
    d_R, d_t = api.DeviceArray((1, 4), np.float64), api.DeviceArray((1, 2), np.float64)
    d_res = api.DeviceArray((1,), api.RESULT_DTYPE)
    d_tr = api.DeviceArray((1, max_iter, 8), np.float64)
    st = api.Stream()
    for _ in range(3):
        icp.fit_batch_from_dev(d_pts, d_off, d_nga, 1, d_R0, d_t0, d_R, d_t, 5.0, d_res, None, st)
    st.synchronize()
    ms = []
    for _ in range(reps):
        a, b = api.Event(), api.Event()
        a.record(st)
        icp.fit_batch_from_dev(d_pts, d_off, d_nga, 1, d_R0, d_t0, d_R, d_t, 5.0, d_res, None, st)
        b.record(st)
        b.synchronize()
        ms.append(a.elapsed_ms(b))
    d_tr.zero()
    icp.fit_batch_from_dev(d_pts, d_off, d_nga, 1, d_R0, d_t0, d_R, d_t, 5.0, d_res, d_tr, st)
    st.synchronize()
    res = d_res.download()[0]
    return np.array(ms), d_R.download()[0], d_t.download()[0], res, d_tr.download()[0]


def stamps(icp):
    L = api.lib()
    if not hasattr(L, "slam_icp_debug_spread_stamps"):
        return None
    buf = np.zeros(256 * 64 * 16, np.int64)
    parts, iters = C.c_int(0), C.c_int(0)
    L.slam_icp_debug_spread_stamps.restype = C.c_int
    L.slam_icp_debug_spread_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if L.slam_icp_debug_spread_stamps(icp.h, buf.ctypes.data, buf.size, C.byref(parts), C.byref(iters)) != 0:
        return None
    return buf[:parts.value * iters.value * 16].reshape(parts.value, iters.value, 16)


def stamp_report(s, n_iter):
    """s: [parts][iters][8] ticks of 10 ns.  Per iteration over the ACTIVE workgroups (those that stamped)."""
    act = s[:, 0, 0] > 0
    raw = s[act][:, :n_iter]
    s = s[act][:, :n_iter].astype(np.float64) * 0.01          # microseconds
    out = {"active_workgroups": int(act.sum())}
    it_start = s[:, :, 0].min(0)
    it_len = np.diff(np.append(it_start, s[:, -1, 5].max()))
    out["us_per_iteration"] = [round(x, 2) for x in it_len]
    # per iteration: the slowest workgroup's search (start -> first barrier), mean search, exchange (barrier -> gathered), solve, second barrier
    search = s[:, :, 2] - s[:, :, 0]
    out["search_us_max"] = [round(x, 2) for x in search.max(0)]
    out["search_us_mean"] = [round(x, 2) for x in search.mean(0)]
    out["search_wave0_us_mean"] = [round(x, 2) for x in (s[:, :, 1] - s[:, :, 0]).mean(0)]
    # from the LAST workgroup through its searches to the first workgroup holding all sums: the exchange proper
    out["exchange_after_last_us"] = [round(x, 2) for x in (s[:, :, 3].min(0) - s[:, :, 2].max(0))]
    out["exchange_spread_us"] = [round(x, 2) for x in (s[:, :, 3].max(0) - s[:, :, 3].min(0))]
    out["solve_us_mean"] = [round(x, 2) for x in (s[:, :, 4] - s[:, :, 3]).mean(0)]
    out["to_next_us_mean"] = [round(x, 2) for x in (s[:, :, 5] - s[:, :, 4]).mean(0)]
    if raw is not None and (raw[:, :, 7] > 0).any():
        out["searches_through_l2_total"] = [int(x) for x in raw[:, :, 7].sum(0)]
        out["searches_through_l2_max_per_workgroup"] = [int(x) for x in raw[:, :, 7].max(0)]
        # the slowest workgroup of every iteration: its searches through L2
        slow = search.argmax(0)
        out["slowest_workgroup_l2_searches"] = [int(raw[slow[k], k, 7]) for k in range(n_iter)]
        nz = raw[:, :, 7] == 0
        out["search_us_max_without_l2"] = [round(float(search[:, k][nz[:, k]].max()), 2) if nz[:, k].any() else None for k in range(n_iter)]
    stage = s[:, :, 6]
    if os.environ.get("SLAM_SPREAD_TILE", "1") not in ("0", "1") and (stage > 0).any():
        k = int(np.argmax((stage > 0).any(0)))
        on = stage[:, k] > 0
        out["first_of_two_stagings_us_max"] = round(float((s[on, k, 1] - s[on, k, 0]).max()), 2)
        out["second_of_two_stagings_us_max"] = round(float((stage[on, k] - s[on, k, 1]).max()), 2)
    if raw.shape[2] > 14 and (raw[:, :, 14] > 0).any():
        # the slowest workgroup of every iteration: its slowest wavefront's pass = tile searches + searches through L2 + the rest
        slow = search.argmax(0)
        out["slowest_wave_us_tile_l2_all"] = [[round(float(raw[slow[k], k, 12 + j]) * 0.01, 2) for j in range(3)] for k in range(n_iter)]
        out["mean_wave_us_tile_l2_all"] = [[round(float(raw[:, k, 12 + j].mean()) * 0.01, 2) for j in range(3)] for k in range(n_iter)]
    if (raw[:, :, 11] > 0).any():
        # every staging: (iteration, tile points, cell-table entries, rows, microseconds), the five slowest and the five largest
        ev = [(int(k), int(raw[w, k, 8]), int(raw[w, k, 9]), int(raw[w, k, 10] & 0xffff), round(float(raw[w, k, 11]) * 0.01, 2))
              for w in range(raw.shape[0]) for k in range(n_iter) if raw[w, k, 11] > 0]
        out["stagings"] = len(ev)
        out["stagings_slowest"] = sorted(ev, key=lambda e: -e[4])[:6]
        out["stagings_largest"] = sorted(ev, key=lambda e: -e[1])[:6]
        out["stagings_median"] = sorted(ev, key=lambda e: e[4])[len(ev) // 2]
    if (raw[:, :, 12] != 0).any():
        # the slowest query of the slowest workgroups, iteration by iteration: [us, how (1 gate square, 2 rows on the index, 4 tile), class, x, y, neighbour]
        q = []
        for k in range(min(n_iter, 4)):
            key = raw[:, k, 12].astype(np.uint64)
            order = np.argsort(-(key >> np.uint64(32)).astype(np.int64))[:4]
            row = []
            for w in order:
                kk = int(key[w])
                xy = np.array([int(raw[w, k, 13]) & 0xffffffff, (int(raw[w, k, 13]) >> 32) & 0xffffffff], np.uint32).view(np.float32)
                row.append([round((kk >> 32) * 0.01, 2), kk & 7, (kk >> 3) & 1, round(float(xy[0]), 3), round(float(xy[1]), 3), int(raw[w, k, 14])])
            q.append(row)
        out["slowest_queries"] = q
    if (stage > 0).any():
        out["stage_us"] = [round(float((stage[:, k][stage[:, k] > 0] - s[:, k, 0][stage[:, k] > 0]).max()), 2) if (stage[:, k] > 0).any() else 0 for k in range(n_iter)]
    return out


def main():
    import oracle_lib as O
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    out = {}
    d = np.load(os.path.join(ROOT, "tests", "golden", "spread_case3.npz"))
    m_ga, m_nga = d["m_ga"].astype(np.float64), d["m_nga"].astype(np.float64)
    kw = {}
    if os.environ.get("SLAM_SPREAD_TILE") == "0":
        kw["spread_tile"] = -1
    elif os.environ.get("SLAM_SPREAD_TILE") is not None:
        kw["spread_tile"] = 1
    kw3 = dict(kw)
    if os.environ.get("SLAM_SPREAD_CELL"):          # the pitch of config 3's index forced (sweeps)
        kw3["cell_size"] = float(os.environ["SLAM_SPREAD_CELL"])
    icp = api.Icp(m_ga, m_nga, **kw3)
    model = O.IcpModel(m_ga, m_nga)
    out["config3_index"] = icp.index_info()
    for k in (1, 5, 9):
        s_ga, s_nga, R0, t0 = d["s_ga%d" % k], d["s_nga%d" % k], d["R%d" % k], d["t%d" % k]
        ms, R, t, res, tr = time_fit(icp, s_ga, s_nga, R0, t0, reps)
        Ro, to, tro, steps = model.fit(s_ga, s_nga, R0.reshape(2, 2), t0, O.icp_params(20, 1e-6, 5.0))
        it = int(res["iters"])
        e = {"scene_points": int(len(s_ga) + len(s_nga)), "iterations": it, "oracle_iterations": int(steps), "n_corr": int(res["n_corr"]),
             "oracle_n_corr": int(tro[steps - 1, 7]),
             "us_per_fit_median": round(float(np.median(ms)) * 1e3, 1), "us_per_fit_min": round(float(ms.min()) * 1e3, 1),
             "us_per_iteration": round(float(np.median(ms)) * 1e3 / max(it, 1), 2),
             "pose_diff_vs_oracle": [float(np.abs(t - to).max()), float(np.abs(R.reshape(2, 2) - Ro).max())]}
        st = stamps(icp) if os.environ.get("SLAM_SPREAD_STAMPS") else None
        if st is not None:
            e["stamps"] = stamp_report(st, it)
        out["config3_cloud%d" % k] = e
    icp.close()
    for name, npts in (("room_2x19999", 39998), ("room_10k", 10000)):
        mg, mn = synth.make_map(npts)
        batch = synth.make_batch(4, n_loop=256)
        icp = api.Icp(mg, mn, **kw)
        model = O.IcpModel(mg, mn)
        es = []
        for s in range(2):
            ga, nga = batch.scan(s)
            ms, R, t, res, tr = time_fit(icp, ga, nga, batch.R[s], batch.t[s], reps)
            Ro, to, tro, steps = model.fit(ga, nga, batch.R[s].reshape(2, 2), batch.t[s], O.icp_params(20, 1e-6, 5.0))
            e = {"iterations": int(res["iters"]), "oracle_iterations": int(steps), "us_per_fit_median": round(float(np.median(ms)) * 1e3, 1),
                 "us_per_iteration": round(float(np.median(ms)) * 1e3 / max(int(res["iters"]), 1), 2),
                 "pose_diff_vs_oracle": [float(np.abs(t - to).max()), float(np.abs(R.reshape(2, 2) - Ro).max())]}
            st = stamps(icp) if os.environ.get("SLAM_SPREAD_STAMPS") else None
            if st is not None and s == 0:
                e["stamps"] = stamp_report(st, int(res["iters"]))
            es.append(e)
        out[name] = {"index": icp.index_info(), "scans": es}
        icp.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
