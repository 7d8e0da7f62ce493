#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X.

Default (BASELINE config 2 per GPU): one "step" = one pass of the per-scan hot path over one batch resident in
HBM: 256 synthetic 1081-beam scans, each registered against the 10k-point map with 30 fixed ICP iterations
(min_delta = -1), then ray-cast from their registered poses into one 2000 x 2000 @ 0.05 m grid (Bresenham free
space + hits), then the counts are folded into the evidence / occupancy planes (finalize).  With N > 1 every rank
does that for its own 256 scans (weak scaling) and the rows of the int32 hit/miss planes that any rank touched are
merged with one RCCL all-reduce per step (slam_grid_merge_begin / _finish of the library) before finalize.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N ...    # starts its own N ranks (the launch below) before touching a GPU; --dry-launch prints it
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config 3      # 64-ring clouds through the CCICP chain, one scan at a time (the reference's usage)
    python bench.py --config 4      # one GPU's share of config 4: 1024 scans, 4000 x 4000 grid
    python bench.py --config 5      # streaming mapper: sliding-window target, periodic merge, PCIe inclusive

Rank 0 prints ONE JSON line.  `value` is registered scan-points/s over the whole job with inputs resident in HBM;
`grid_cell_updates_per_s` is the second half of BASELINE.json's metric; `value_pcie_inclusive` is the same workload
fed from pinned host memory through the streaming mapper (SURVEY 8(d) metric (1): H2D of scans + kernels + D2H of
poses); `model_build_ms` is what 8(d) asks to report separately (the map upload + index build of a match).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime deals a process's streams over 4 hardware queues unless told otherwise, and two streams on one queue run
# one after the other: this process holds up to eight that must overlap (registration x 2, grid, copy, index build, RCCL's).
# Read by the runtime when it initialises: set before anything touches HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

N_SCANS, N_ITERS, GRID, RES, MAP_POINTS = 256, 30, 2000, 0.05, 10000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_JSON_FD = None
_RK = None       # slam_amd.ranks.Ranks of an N > 1 run: a failure anywhere is announced through it (main guard at the bottom)


def quiet_stdout():
    """From here on file descriptor 1 is stderr: gloo, RCCL and the HIP runtime print banners on stdout from C++ ("[Gloo] Rank 0
    is connected to ...", the RCCL version line), and stdout must carry ONE JSON line.  emit() writes to the saved descriptor."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, line)


def host_cpus():
    """What this process may use of the host: logical CPUs, physical cores (distinct (package, core) pairs of /proc/cpuinfo), the
    CPUs of its affinity mask, and its cgroup's CPU quota in CPUs (cpu.max of cgroup v2 / cfs_quota of v1; None = unlimited).  A GPU
    box of this pool shows all 256 logical CPUs of a two-socket host and grants a one-GPU job 16 of them through the quota."""
    logical = os.cpu_count() or 1
    phys = set()
    try:
        pkg = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if pkg is not None and core is not None:
                    phys.add((pkg, core))
                pkg = core = None
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = logical
    return {"logical": logical, "physical": len(phys) or logical, "affinity": affinity, "quota": quota}


def cpu_baseline(m_ga, m_nga, batch, grid_size, res, p2l=False):
    """Times the CPU oracle (a port of the reference path; tests pin it) on this host: ICP over the WHOLE batch with OpenMP over
    scans on every core this process is granted, plus the Bresenham update of those scans.  Test infrastructure used as a reported
    baseline only -- never on the measured path.  `cores` = the threads used: the host's physical cores, or the cgroup's CPU quota
    where that is smaller (more threads than granted CPUs only get throttled)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    hc = host_cpus()
    grant = min(hc["physical"], hc["affinity"])
    if hc["quota"]:
        grant = min(grant, max(1, int(hc["quota"] + 0.5)))
    threads = max(1, grant)
    # every thread gets at least four scans (dynamic schedule over scans: two per thread leaves the last ones alone at the end)
    copies = max(1, -(-4 * threads // batch.n_scans))
    sub = batch
    pts = np.concatenate([batch.pts] * copies) if copies > 1 else batch.pts
    off = np.concatenate([[0]] + [batch.scan_off[1:] + k * batch.n_points for k in range(copies)]).astype(np.int32) if copies > 1 else batch.scan_off
    nga = np.tile(batch.scan_nga, copies)
    R0, t0 = np.tile(batch.R, (copies, 1)), np.tile(batch.t, (copies, 1))
    model = O.IcpModel(m_ga, m_nga, normals_k=10 if p2l else 0)
    p = O.icp_params(N_ITERS, -1.0, 5.0, O.NN_KDTREE, O.MODE_P2L if p2l else O.MODE_P2P)
    model.fit_batch(pts, off, nga, R0, t0, p, n_threads=threads)      # threads started, scratch allocated, pages touched
    # repeated until about 20 core-seconds (and at least 2 s of wall time) have gone into it
    reps, t_icp = 0, 0.0
    while (t_icp * threads < 20.0 or t_icp < 2.0) and reps < 400 and t_icp < 30.0:
        t_a = time.perf_counter()
        R, t, iters, ncorr, delta = model.fit_batch(pts, off, nga, R0, t0, p, n_threads=threads)
        t_icp += time.perf_counter() - t_a
        reps += 1
    t_icp /= reps
    n_pts_icp = batch.n_points * copies
    R, t = R[:batch.n_scans], t[:batch.n_scans]
    # single-thread rate on a sixteenth of the batch (the reference's own execution model)
    one = sub.shard(0, max(1, sub.n_scans // 16))
    t_a = time.perf_counter()
    model.fit_batch(one.pts, one.scan_off, one.scan_nga, one.R, one.t, p, n_threads=1)
    t_icp1 = time.perf_counter() - t_a
    g = O.grid_params(grid_size, grid_size, res, min_cluster_points=20)
    hits = np.zeros(grid_size * grid_size, np.int32)
    misses = np.zeros(grid_size * grid_size, np.int32)
    ends, origins = [], []
    for s in range(sub.n_scans):
        o, e = sub.scan_off[s], sub.scan_off[s + 1]
        ends.append(O.transform_points(sub.pts[o:e], R[s], t[s]))
        origins.append(np.tile(t[s].astype(np.float32), (e - o, 1)))
    ends, origins = np.concatenate(ends), np.concatenate(origins)
    t_a = time.perf_counter()
    _, _, upd = O.grid_raycast(g, origins, ends, hits, misses, n_threads=threads)
    t_grid = time.perf_counter() - t_a
    t_a = time.perf_counter()
    n1 = len(ends) // 8
    _, _, upd1 = O.grid_raycast(g, origins[:n1], ends[:n1], hits, misses)
    t_grid1 = time.perf_counter() - t_a
    # contended atomics can make the threaded grid update slower than one thread: the baseline takes the faster
    t_grid_best = min(t_grid, t_grid1 * upd / max(upd1, 1))
    t_icp_batch = t_icp / copies                       # ICP time of ONE batch's worth of scans
    return {
        "value": sub.n_points / (t_icp_batch + t_grid_best), "unit": "points/s", "cores": threads,
        "kind": "port",
        "solver": "point-to-line (oracle fit_step_p2l: icpPointToPlane.cpp:37-107)" if p2l else "point-to-point (icpPointToPoint.cpp:33-172)",
        "sample": "%d x (%d scans = %d cop%s of the %d-scan batch x %d ICP iterations; kd-tree NN, OpenMP over scans, %d threads, "
                  "-O3) + Bresenham of one batch's scans into the %dx%d grid (the faster of OpenMP over beams with atomic "
                  "increments and one thread)"
                  % (reps, batch.n_scans * copies, copies, "y" if copies == 1 else "ies", batch.n_scans, N_ITERS, threads, grid_size, grid_size),
        "icp_points_per_s": n_pts_icp / t_icp,
        "icp_points_per_s_1thread": one.n_points / t_icp1,
        "icp_speedup_over_1thread": (n_pts_icp / t_icp) / (one.n_points / t_icp1),
        "grid_cell_updates_per_s": upd / t_grid,
        "grid_cell_updates_per_s_1thread": upd1 / t_grid1,
        "host_cores": hc["logical"], "host_physical_cores": hc["physical"], "cpu_affinity": hc["affinity"],
        "cpu_quota_cpus": hc["quota"],
        "cores_means": "threads used = min(physical cores, affinity mask, the cgroup's CPU quota): what the host grants this job",
    }


def rccl_channels():
    """What RCCL's own INIT log (NCCL_DEBUG_FILE, set in main() on rank 0) says about the communicator: collective channels =
    the workgroups one of its all-reduce kernels occupies.  None when the log is absent or says nothing recognisable."""
    import re
    f = os.environ.get("NCCL_DEBUG_FILE")
    if not f or not os.path.exists(f):
        return None
    try:
        txt = open(f, errors="replace").read()
    except Exception:
        return None
    m = re.findall(r"(\d+) coll channels", txt)
    if m:
        return int(m[-1])
    m = re.findall(r"Channel \d+/(\d+)", txt)
    return int(m[-1]) if m else None


def pmc_profile(p2l=False):
    """The committed PMC summary of this same command (profiles/rNN_traffic.json, rNN_p2l_traffic.json for --mode p2l;
    written by tools/summarize_profiles.py from separate rocprofv3 --pmc passes): (file name, dict) or (None, {})."""
    import re
    pat = re.compile(r"^r\d+_p2l_traffic\.json$" if p2l else r"^r\d+_traffic\.json$")
    files = sorted(os.path.join(ROOT, "profiles", f) for f in os.listdir(os.path.join(ROOT, "profiles")) if pat.match(f))
    if not files:
        return None, {}
    try:
        return os.path.relpath(files[-1], ROOT), json.load(open(files[-1]))
    except Exception:
        return None, {}


def csrc_sha():
    """sha256 over the kernel sources this run was built from (tools/summarize_profiles.py stamps the committed PMC summary with
    the same hash: a summary taken from other sources is reported as stale)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def model_build_times(api, synth, m_ga, m_nga, **icp_kw):
    """slam_icp_create (Icp::Icp, icp.cpp:26-70: model copy + index) wall time per call, the buffer pool warm:
    the 10 k-point map of the bench and a 2 x 19 999-point model (the CCICP cap, icpTools.h:21)."""
    rs = np.random.RandomState(3)
    big = (rs.randn(19999, 2) * [30.0, 20.0], rs.rand(19999, 2) * [80.0, 60.0] - [40.0, 30.0])
    out = {}
    for name, (ga, nga) in (("map_%d_points" % (len(m_ga) + len(m_nga)), (m_ga, m_nga)), ("model_2x19999_points", big)):
        api.Icp(ga, nga, **icp_kw).close()
        ts, parts = [], None
        for _ in range(12):
            t0 = time.perf_counter()
            icp = api.Icp(ga, nga, **icp_kw)
            ts.append(time.perf_counter() - t0)
            parts = icp.build_info()
            icp.close()
        out[name] = {"ms": float(np.median(ts) * 1e3), "min_ms": float(min(ts) * 1e3), "on_device": parts[0],
                     "host_ms_enqueue|the_one_wait": [round(x, 4) for x in parts[1][:2]]}
    return out


def single_scan_times(api, synth, m_ga, m_nga, **icp_kw):
    """BASELINE config 1 on the GPU -- one 1081-beam scan against the 10 k-point map through the host API
    (slam_icp_fit: Icp::fit of one doICPMatch, icpTools.cpp:188), 20 iterations."""
    batch = synth.make_batch(1, n_loop=256)
    t_ga, t_nga = batch.scan(0)
    icp = api.Icp(m_ga, m_nga, max_iter=20, min_delta=-1.0, **icp_kw)
    icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        icp.fit(t_ga, t_nga, batch.R[0], batch.t[0], 5.0)
        ts.append(time.perf_counter() - t0)
    icp.close()
    n = len(t_ga) + len(t_nga)
    return {"fit_ms_1081_points_20_iterations_host_api": float(np.median(ts) * 1e3),
            "registered_points_per_s": n / float(np.median(ts))}


def stream_rate(api, synth, m_ga, m_nga, batch, grid_size, n_chunks=48, **kw):
    """The same batch fed from pinned host memory through the streaming mapper (H2D | ICP | raycast on three streams):
    seconds per chunk in steady state, PCIe inclusive."""
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20, raycast_wg_per_cu=kw.pop("raycast_wg_per_cu", 0)),
                    grid_size_x=grid_size, grid_size_y=grid_size,
                    resolution=RES, max_scans=batch.n_scans, max_points=batch.n_points,
                    icp=dict(max_iter=N_ITERS, min_delta=-1.0), **kw)
    for s in [mp.push(batch) for _ in range(mp.n_slots)]:   # warm-up: every slot, every scratch buffer
        mp.wait(s)
    api.synchronize()
    t0 = time.perf_counter()
    pending = []
    for k in range(n_chunks):
        if len(pending) == mp.n_slots:
            mp.wait(pending.pop(0))          # poses of the oldest chunk in flight back on the host, its slot free
        pending.append(mp.push(batch))
    for s in pending:
        R, t = mp.wait(s)
    mp.finish()
    dt = (time.perf_counter() - t0) / n_chunks
    st = mp.stats()
    mp.close()
    return dt, R, t, st


def endpoint_inputs(api, synth, batch, R, t, at_size=True):
    """The two inputs of the endpoint leg, as float32 xyz rows: (name, obstacle set, ground set).
    config 2: every registered scan point as an obstacle endpoint in the map frame ((float)(R p + t), icpPointToPoint.cpp:69-70's
    rounding); config 3: one 64-ring cloud split by the ground segmentation into its drv (obstacle below the robot's height)
    and ground sets, as MLS::addToOccupancy does it (mls.cpp:61-67)."""
    ends = []
    for s in range(batch.n_scans):
        p = batch.pts[batch.scan_off[s]:batch.scan_off[s + 1]]
        Rs, ts = R[s].reshape(4), t[s]
        ends.append(np.stack([(Rs[0] * p[:, 0] + Rs[1] * p[:, 1]) + ts[0], (Rs[2] * p[:, 0] + Rs[3] * p[:, 1]) + ts[1]], 1).astype(np.float32))
    ends = np.concatenate(ends)
    ends = np.concatenate([ends, np.zeros((len(ends), 1), np.float32)], 1)
    xyz = synth.make_cloud3d(3, n_loop=50)[0]
    seg = api.GroundSegmentation()
    lab = seg.segment(xyz)
    seg.close()
    # ... and the kernel where neither the launch nor one hot wall bounds it: 8 M points spread over the grid's 100 m x 100 m
    rs = np.random.RandomState(77)
    wide = np.concatenate([rs.uniform(-49.9, 49.9, (8_000_000, 2)), np.zeros((8_000_000, 1))], 1).astype(np.float32)
    return [("config 2: the %d registered endpoints of the batch as obstacle points" % len(ends), ends, np.zeros((0, 3), np.float32))] + \
           ([("at size: 8 000 000 obstacle points spread uniformly over the grid (no hot wall, launch latency amortised)", wide,
              np.zeros((0, 3), np.float32))] if at_size else []) + [
            ("config 3: one 64-ring cloud, segmented: %d drv (obstacle) + %d ground points" %
             (int((lab == api.GSEG_OBSTACLE).sum()), int((lab == api.GSEG_GROUND).sum())),
             np.ascontiguousarray(xyz[lab == api.GSEG_OBSTACLE]), np.ascontiguousarray(xyz[lab == api.GSEG_GROUND]))]


def endpoint_leg(api, synth, batch, R, t, grid_size, res, with_cpu, at_size=True):
    """The grid update the reference actually performs -- MLS::addToOccupancy's endpoint binning (mls.cpp:73-142: one counter
    per accepted point, no free-space traversal) -- through slam_grid_add_endpoints_dev + slam_grid_finalize_reset.
    SURVEY 8(d): 16 B per cell update (8 B point xy + 8 B counter read-modify-write) against the ~1.3 TB/s ceiling of global
    atomics.  The CPU oracle's loop (ogrid_add_endpoints, one thread: the reference's execution model) is timed beside it."""
    out, reps = [], 40
    st = api.Stream()
    for name, obs, gnd in endpoint_inputs(api, synth, batch, R, t, at_size):
        g = api.Grid(grid_size, grid_size, res, rolling=0, min_cluster_points=20)
        d_obs = api.DeviceArray.from_host(obs if len(obs) else np.zeros((1, 3), np.float32))
        d_gnd = api.DeviceArray.from_host(gnd if len(gnd) else np.zeros((1, 3), np.float32))

        def add():
            api.check(api.lib().slam_grid_add_endpoints_dev(g.h, d_obs.ptr, len(obs), d_gnd.ptr, len(gnd), 3, st.ptr))
        for _ in range(3):
            add()
            g.finalize_reset(st)
        st.synchronize()
        u0 = g.total_updates()
        add()
        st.synchronize()
        upd = g.total_updates() - u0
        g.finalize_reset(st)
        # replayed from hipGraphs: launched call by call from Python the host's enqueue rate (15-25 us per call) is what an
        # event pair would measure, not these microsecond kernels
        st.synchronize()
        g_k, g_pair = api.Graph(st), api.Graph(st)
        with g_k:
            for _ in range(reps):                # the kernel alone, back to back (counts accumulate; one fold after)
                add()
        with g_pair:
            for _ in range(reps):                # the update as a mapper issues it: endpoints, then finalize + count reset
                add()                            # (captured, slam_grid_finalize_reset is its two-call form: grid.hip)
                g.finalize_reset(st)
        e = [api.Event() for _ in range(4)]
        e[0].record(st)
        g_k.launch()
        e[1].record(st)
        g.finalize_reset(st)
        e[2].record(st)
        g_pair.launch()
        e[3].record(st)
        st.synchronize()
        ms_k = e[0].elapsed_ms(e[1]) / reps
        ms_pair = e[2].elapsed_ms(e[3]) / reps
        leg = {"input": name, "grid": [grid_size, grid_size], "cell_updates_per_call": int(upd),
               "endpoints_kernel_ms": ms_k, "endpoints_plus_finalize_reset_ms": ms_pair,
               "cell_updates_per_s": upd / (ms_pair * 1e-3), "cell_updates_per_s_kernel_alone": upd / (ms_k * 1e-3),
               "alg_bytes": 16 * int(upd), "GBps": 16 * upd / (ms_k * 1e-3) / 1e9,
               "vs_global_atomic_ceiling": 16 * upd / (ms_k * 1e-3) / 1e9 / 1300.0}
        if with_cpu:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            gp = O.grid_params(grid_size, grid_size, res, min_cluster_points=20)
            hits, misses = np.zeros(grid_size * grid_size, np.int32), np.zeros(grid_size * grid_size, np.int32)
            O.grid_add_endpoints(gp, obs, gnd, hits, misses)
            t0 = time.perf_counter()
            for _ in range(5):
                O.grid_add_endpoints(gp, obs, gnd, hits, misses)
            leg["cpu_oracle_cell_updates_per_s_1thread"] = 5 * upd / (time.perf_counter() - t0)
        out.append(leg)
        g.close()
    return {"what": "MLS::addToOccupancy's endpoint update (mls.cpp:73-142) via slam_grid_add_endpoints_dev + slam_grid_finalize_reset; "
                    "alg_bytes = 16 B per accepted point (SURVEY 8(d)), GBps over the endpoints kernel's own time, ceiling = 1.3 TB/s of "
                    "global atomics (MI355X_MICROARCH.md)", "legs": out}


def config3_oracle_beside(detail, n_clouds):
    """What the truth errors of the config-3 leg mean: the CPU oracle's CCICP chain (tests/ccicp_chain.py: ground segmentation,
    GA/NGA classes, 0.5 m voxel centroids, crop + split, kd-tree ICP; icpTools.cpp:222-298 from oracle pieces) on the SAME
    matches of the C++ adapter's sequence -- same target cloud, same scene cloud, same initial pose -- its error against the
    truth beside the GPU's, and the two poses' difference.  Checker code, used here as a reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from ccicp_chain import oracle_scan_match
    from slam_amd import synth
    t0 = time.perf_counter()
    clouds = [synth.make_cloud3d(k, n_loop=50)[0] for k in range(n_clouds)]
    seg = {}
    e_gpu, e_or, dxy, dyaw = [], [], [], []
    for m, j in enumerate(detail["target_of"]):
        if j not in seg:
            lab = O.gseg_segment(clouds[j])[0]
            seg[j] = (clouds[j][lab >= O.GSEG_OBSTACLE], clouds[j][lab == O.GSEG_GROUND])
        e = oracle_scan_match(seg[j][0], seg[j][1], clouds[m + 1], list(detail["init"][m]))
        g, tr = detail["poses"][m], detail["truth"][m]
        e_gpu.append(float(np.hypot(g[0] - tr[0], g[1] - tr[1])))
        e_or.append(float(np.hypot(e["t"][0] - tr[0], e["t"][1] - tr[1])))
        dxy.append(float(np.hypot(g[0] - e["t"][0], g[1] - e["t"][1])))
        d = 2.0 * np.arctan2(g[5], g[6]) - e["yaw"]
        dyaw.append(float(abs((d + np.pi) % (2 * np.pi) - np.pi)))
    return {"matches": len(e_or), "gpu_mean_xy_error_m": float(np.mean(e_gpu)), "gpu_max_xy_error_m": float(np.max(e_gpu)),
            "oracle_mean_xy_error_m": float(np.mean(e_or)), "oracle_max_xy_error_m": float(np.max(e_or)),
            "max_pose_difference_gpu_vs_oracle": {"xy_m": float(np.max(dxy)), "yaw_rad": float(np.max(dyaw))},
            "oracle_seconds": time.perf_counter() - t0,
            "meaning": "the error against the truth is the CHAIN's (0.5 m voxel centroids of a 64-ring cloud, a few hundred correspondences, "
                       "a target up to ten poses back), identical for the CPU oracle and the GPU to the tolerance of north_star "
                       "(1e-4 m / 1e-5 rad): see max_pose_difference_gpu_vs_oracle"}


def run_config3(n_clouds, with_oracle=False):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_config3
    out = bench_config3.measure(n_clouds)
    detail = out.pop("_cpp_detail", None)
    if with_oracle and detail is not None:
        try:
            out["cpp_adapter"]["oracle_beside"] = config3_oracle_beside(detail, n_clouds)
        except Exception as ex:
            out["cpp_adapter"]["oracle_beside"] = {"error": repr(ex)}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n_ranks, argv, port=None):
    """The launch the driver makes for N > 1 (one rank per GPU over RCCL), built here when bench.py is started by itself."""
    argv = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + argv


def visible_devices():
    """GPUs this process could use, counted without initialising HIP (torch.cuda.device_count() does not, on this
    image): the launcher itself must stay off the GPU."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception as ex:
        log("could not count the devices (%s)" % ex)
        return -1


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `python -m torch.distributed.run ...
    bench.py --gpus N ...` BEFORE this process makes any HIP call, hand its one JSON line on, and fail loudly (the
    child's stderr tail, a non-zero exit) when it fails.  Returns the exit code."""
    import subprocess
    cmd = launch_command(args.gpus, argv)
    if args.dry_launch:
        print(json.dumps({"launch": cmd, "n_ranks": args.gpus, "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}}), flush=True)
        return 0
    if not args.one_device:
        n_dev = visible_devices()
        if 0 <= n_dev < args.gpus:
            log("bench.py --gpus %d: %d ranks need %d devices, this machine shows %d (one process per GPU over RCCL; "
                "--backend gloo --one-device rehearses the N > 1 path on one GPU)" % (args.gpus, args.gpus, args.gpus, n_dev))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL between processes needs it on this driver
    log("launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    out, err = p.communicate()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        log("the %d-rank launch failed (exit code %d); the end of its stderr:" % (args.gpus, p.returncode))
        log(err[-4000:])
        return p.returncode or 1
    sys.stderr.write(err[-2000:])
    print(lines[-1], flush=True)
    return 0


def side_run(argv, keys, timeout=420):
    """One of the other configs as a child `python bench.py ...` (never under the profiler's exec rules: a plain child);
    returns the chosen keys of its JSON line plus its workload, or {"error": ...} -- the headline must not depend on it."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": "exit code %d: %s" % (p.returncode, p.stderr[-400:])}
        d = json.loads(lines[-1])
        r = {k: d.get(k) for k in keys}
        r["workload"] = d.get("config", {}).get("workload")
        r["command"] = "python bench.py " + " ".join(argv)
        return r
    except Exception as ex:
        return {"error": repr(ex)}


def main():
    global GRID
    # dmabuf IPC between the ranks' processes (RCCL needs it on this driver); already exported on the boxes, set here for a launcher
    # that starts the ranks with an environment of its own.  Before anything loads the HIP runtime.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE config: 2 (default, the headline), 3 (64-ring clouds, one match per cloud), 4 (one GPU's "
                         "share: 1024 scans, 4000^2 grid), 5 (streaming mapper)")
    ap.add_argument("--scans", type=int, default=None, help="scans per GPU (default: 256; config 4: 1024)")
    ap.add_argument("--grid", type=int, default=None, help="grid side in cells (default: 2000; config 4: 4000)")
    ap.add_argument("--map-points", type=int, default=MAP_POINTS,
                    help="model points of the registration target (default 10 000: BASELINE configs 1/2/4).  The reference's own cap is "
                         "2 x 19 999 (icpTools.h:21, icpTools.cpp:255-274): --map-points 39998")
    ap.add_argument("--map-kind", choices=["room", "uniform"], default="room",
                    help="room = the synthetic room's walls and pillars sampled more densely (SURVEY 8(d)); uniform = half the points per "
                         "class uniformly random over the room's 40 m x 30 m (no structure for the index to exploit, and nothing to register "
                         "against: timing only, the pose check is skipped)")
    ap.add_argument("--wave-tiles", type=int, default=0, help="slam_icp_params::wave_tiles (1 = per-wavefront model tiles in LDS for models too large for LDS; 0 = library default, off)")
    ap.add_argument("--list-min-halo", type=float, default=0.0, help="slam_icp_params::list_min_halo in metres (0 = library default, < 0 = the finest lattice that fits)")
    ap.add_argument("--clouds", type=int, default=50, help="config 3: clouds of the sequence")
    ap.add_argument("--stream-scans", type=int, default=10240, help="config 5: scans of the stream per GPU")
    ap.add_argument("--chunk", type=int, default=256, help="config 5: scans per chunk")
    ap.add_argument("--window", type=int, default=4, help="config 5: chunks in the sliding local map (0 = fixed target)")
    ap.add_argument("--thin", type=float, default=0.1, help="config 5: pitch of the window's thinning lattice in metres (0 = stride)")
    ap.add_argument("--rebuild-every", type=int, default=4, help="config 5: chunks between rebuilds of the target")
    ap.add_argument("--merge-every", type=int, default=8, help="config 5: chunks between merges over the GPUs + finalize")
    ap.add_argument("--reg-streams", type=int, default=0, help="config 5: slam_mapper_params::registration_streams (0 = library default)")
    ap.add_argument("--slots", type=int, default=0, help="config 5: slam_mapper_params::slots (0 = library default)")
    ap.add_argument("--pair-scans", type=int, default=0, help="config 5: slam_icp_params::pair_scans of the mapper's registrations (0 = library default)")
    ap.add_argument("--mode", choices=["p2p", "p2l"], default="p2p",
                    help="the ICP step: p2p = what the reference compiles and calls (icpPointToPoint.cpp:33-172); p2l = the solver "
                         "north_star names, point-to-line error + 3x3 normal equations (icpPointToPlane.cpp:37-107, stale upstream)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per scan point (0 = library default)")
    ap.add_argument("--cell", type=float, default=0.0, help="ICP cell pitch in metres (0 = library default)")
    ap.add_argument("--raycast", choices=["tiled", "merge", "global"], default="tiled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip model build, single scan, streaming and config-3 legs")
    ap.add_argument("--raycast-seg", type=int, default=0, help="slam_grid_params::raycast_seg_items (0 = library default)")
    ap.add_argument("--raycast-wg", type=int, default=None,
                    help="slam_grid_params::raycast_wg_per_cu (0 = library default, two per CU: the fastest raycast call by itself).  "
                         "Not given: 1 for config 2's pipelined step -- beside two registration launches half as many private copies of a "
                         "hot tile are written back, which is worth more than the call's own 20 us: 50 steps, three runs each, point-to-"
                         "point 0.3080 -> 0.3049 ms per step, point-to-line 0.3216 -> 0.3081 (its normals come from L2, which the "
                         "write-backs sweep; tools/exp/rc_sweep.sh) -- and the library default for configs 4 and 5 and the streaming "
                         "mapper, where it makes no difference (tools/exp/rc_sweep2.sh)")
    ap.add_argument("--raycast-max-wg", type=int, default=0, help="slam_grid_params::raycast_max_workgroups (0 = no cap)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the grid update of step k-1 with the registration of step k (one stream: a captured "
                         "hipGraph per step at N=1, call by call otherwise)")
    ap.add_argument("--one-grid", action="store_true",
                    help="pipelined launch: every step into the same grid buffer on one grid stream (default: consecutive steps alternate "
                         "over two grid buffers and two grid streams)")
    ap.add_argument("--no-graph", action="store_true",
                    help="with --no-pipeline at N=1: launch every step call by call instead of replaying one captured hipGraph")
    ap.add_argument("--no-torch", action="store_true", help="N=1 only: do not import torch at all")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 data plane: nccl = the library's merge (slam_grid_merge_begin/_finish) over RCCL; gloo = the same "
                         "entry points over the library's host-staged communicator with gloo carrying the host buffers, "
                         "only to rehearse the N>1 path with several ranks on ONE GPU (--one-device), where RCCL cannot run")
    ap.add_argument("--one-device", action="store_true", help="all ranks use GPU 0 (rehearsal with --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the N>1 code path (communicator + row merge) with one rank")
    ap.add_argument("--no-merge", action="store_true",
                    help="N>1 without the exchange step: every rank runs the N=1 step on its own shard and nothing is summed -- the "
                         "scaling of independent ranks, to set beside the default N>1 run (its difference is the cost of the merge)")
    ap.add_argument("--merge-order", choices=["early", "late"], default=None,
                    help="N>1 pipelined: late (default) = a step's raycast, merge_begin and merge_finish are enqueued together two "
                         "registrations later; early = raycast and merge_begin with the step's own registration (they wait for it on the "
                         "device), merge_finish two registrations later.  With ONE rank (--force-dist, 100 steps, three runs each): "
                         "late 0.363-0.367 ms per step, early 0.373-0.375, no merge 0.343-0.345 -- an early raycast takes CUs from the "
                         "registration that has just started; whether hiding the ranks' exchange behind it pays with 8 ranks is for "
                         "the first multi-GPU lease to say")
    ap.add_argument("--step-streams", type=int, default=2, help="registration streams of the pipelined step (consecutive steps in turn)")
    ap.add_argument("--private-queues", action="store_true",
                    help="the pipelined step's streams each on a hardware queue of their own (CU-masked streams naming every CU)")
    ap.add_argument("--grid-lag", type=int, default=None,
                    help="pipelined launch: the host enqueues the grid update of step k - LAG after the registration of step k (with N>1 that "
                         "call waits for the united row range of step k - LAG: a larger lag is more registrations queued while it waits; 100 steps, "
                         "two runs each: lag 2 0.3374 ms per step, 3 0.3353; one rank over RCCL 0.3626 / 0.3600 / 0.3605 for 2 / 3 / 4)")
    ap.add_argument("--reg-cu-cap", type=int, default=None, metavar="K",
                    help="registration streams leave K CUs of every XCD alone (hipExtStreamCreateWithCUMask), so that the short kernels of "
                         "the other streams -- RCCL's all-reduce, the grid update -- find a CU while 0.6 ms registration workgroups hold "
                         "the rest (0 = ordinary streams)")
    ap.add_argument("--calibrate", action="store_true",
                    help="N>1 over RCCL: before the warm-up, time a few steps with each of --reg-cu-cap 0/1 x --merge-order late/early "
                         "(whichever of the two was not given), take the maximum over the ranks and keep the fastest; reported as "
                         "merge.calibration.  On by default with more than one rank (no multi-GPU lease has measured which setting RCCL's "
                         "kernels need beside 0.6 ms registration workgroups); with --force-dist only when asked for")
    ap.add_argument("--test-kill-rank", type=int, default=-1,
                    help="tests only: this rank ends with SIGKILL behind its warm-up (the others must notice and exit: slam_amd.ranks)")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="N>1 over RCCL: skip the calibration (about 1 700 steps before the warm-up) and run with the one-GPU defaults; recorded "
                         "in the line as config.calibration_skipped, so that a run that was cut short can be told from one that hung")
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of `steps` steps each, run back to back behind the one warm-up; the MEDIAN region is reported "
                         "(ms_per_step_runs lists them all)")
    ap.add_argument("--start-stagger-us", type=float, default=125.0,
                    help="pipelined launch: the host waits this long between enqueueing the first registration of a run and the second "
                         "(on the other registration stream), so that the two streams are out of step from the start as they are in "
                         "the steady state -- inside the timed region, like everything the host does there.  The driver's own command (20 steps, "
                         "5 warm-up), five runs each, median ms per step: 0 us 0.368, 100 0.359, 150 0.359, 200 0.360, 250 0.364, 300 0.366, "
                         "400 0.372 (tools/exp/start_sweep.sh); over 100 steps it is within the noise")
    ap.add_argument("--merge-thread", type=int, default=1, choices=[0, 1],
                    help="N>1 pipelined: 1 (default) = slam_grid_merge_async: the merge of a step (key, the wait for the united range, the "
                         "rows' all-reduce, finalize) is issued by the communicator's helper thread, the enqueue thread never waits for the "
                         "device; 0 = slam_grid_merge_begin/_finish on the enqueue thread (rounds 2-4: it slept 0.26 of every 0.31 ms there)")
    ap.add_argument("--merge-timeout", type=float, default=30.0, help="N>1: seconds after which a merge whose united range has not arrived fails (SLAM_E_TIMEOUT)")
    ap.add_argument("--rank-timeout", type=float, default=120.0, help="N>1: time-out of every barrier / control-plane collective, seconds")
    ap.add_argument("--dead-after", type=float, default=15.0, help="N>1: a rank whose heartbeat has not moved for this long is lost: the others exit")
    ap.add_argument("--no-rccl-log", action="store_true", help="N>1: do not ask RCCL for its INIT log (merge.rccl_channels stays null)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="print the launch `python bench.py --gpus N` would make of its N ranks (one JSON line) and exit")
    args = ap.parse_args()
    if args.raycast_wg is None:
        args.raycast_wg = 1 if (args.config == 2 and not args.no_pipeline) else 0
    S = args.scans or (1024 if args.config == 4 else N_SCANS)
    GRID = args.grid or (4000 if args.config == 4 else GRID)

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch):
        # `python bench.py --gpus N` by itself: this process becomes the launcher of N ranks and touches no GPU
        raise SystemExit(self_launch(args, sys.argv[1:]))

    quiet_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    if args.config == 3:
        assert world == 1, "config 3 is a single-GPU sequence"
        from slam_amd import api
        api.set_device(0)
        out = run_config3(args.clouds, not args.no_cpu_baseline)
        out.update({"n_gpus": 1, "higher_is_better": True, "data": "synthetic", "vs_baseline": None,
                    "dtype": "f64 pose / f32 distance", "device": api.device_info()[0]})
        emit(out)
        return

    torch = dist = None
    if world > 1 or args.force_dist or not args.no_torch:
        # torch first, so that this process runs ONE HIP runtime (torch's) for
        # both the library's kernels and RCCL
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
    from slam_amd import api, synth
    api.set_device(local_rank)
    multi = world > 1 or args.force_dist
    merging = multi and not args.no_merge
    comm = rk = None
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # control plane (rendezvous, barriers with a time-out, the maximum over the ranks' clocks, a watchdog that ends this rank
        # when another is lost: slam_amd/ranks.py) over gloo; the data plane -- the planes' rows -- goes through the library's own
        # RCCL communicator, whose id rank 0 hands out here
        global _RK
        from slam_amd import ranks
        if merging and args.backend == "nccl" and rank == 0 and "NCCL_DEBUG" not in os.environ and not args.no_rccl_log:
            # what RCCL chose for this communicator (channels = workgroups per collective kernel): its own INIT log, to a file
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ["NCCL_DEBUG_SUBSYS"] = "INIT"
            os.environ["NCCL_DEBUG_FILE"] = os.path.join(os.environ.get("TMPDIR", "/tmp"), "slam_rccl_init_%d.log" % os.getpid())
        _RK = rk = ranks.Ranks(timeout_s=args.rank_timeout, dead_after_s=args.dead_after)
        if merging and args.backend == "gloo":
            # rehearsal (several ranks on ONE GPU, where RCCL cannot run): the library's own merge entry points over its
            # host-staged communicator (slam_comm_create_host), gloo carrying the host buffers
            def gloo_allreduce(a, op):
                rk.dist.all_reduce(torch.from_numpy(a), op=rk.dist.ReduceOp.SUM if op == api.COMM_SUM else rk.dist.ReduceOp.MIN)
            comm = api.Comm.host(rank, world, gloo_allreduce)
        if merging and args.backend == "nccl":
            comm = api.Comm(rk.broadcast_object(api.Comm.unique_id() if rank == 0 else None), rank, world)
            # (RCCL's version banner goes where quiet_stdout() sent descriptor 1)
        if comm is not None:
            comm.set_timeout(args.merge_timeout)     # a united range that takes longer is a lost rank: SLAM_E_TIMEOUT, not a hang

    if args.map_kind == "uniform":
        rs_ = np.random.RandomState(4242)
        half = args.map_points // 2
        m_ga = rs_.rand(half, 2) * [synth.ROOM_W, synth.ROOM_H] - [synth.ROOM_W / 2, synth.ROOM_H / 2]
        m_nga = rs_.rand(args.map_points - half, 2) * [synth.ROOM_W, synth.ROOM_H] - [synth.ROOM_W / 2, synth.ROOM_H / 2]
    else:
        m_ga, m_nga = synth.make_map(args.map_points)

    def barrier():
        if multi:
            rk.barrier()

    def sync():
        if comm is not None:
            comm.drain()             # merges still with the communicator's helper thread are enqueued first
        api.synchronize()
        if torch is not None:
            torch.cuda.synchronize()

    if args.config == 5:
        run_config5(args, api, synth, m_ga, m_nga, rank, world, comm, rk if multi else None, sync, barrier)
        return

    # ---- synthetic inputs of BASELINE config 2 (per rank: its own 256 scans of the loop)
    batch = synth.make_batch(S, n_loop=S * world, first=rank * S)
    P = batch.n_points
    p2l = args.mode == "p2l"
    mode_kw = dict(mode=api.ICP_P2L, normals_k=10) if p2l else {}
    if args.wave_tiles:
        mode_kw["wave_tiles"] = args.wave_tiles
    if args.list_min_halo:
        mode_kw["list_min_halo"] = args.list_min_halo
    icp = api.Icp(m_ga, m_nga, max_iter=N_ITERS, min_delta=-1.0, lanes_per_point=args.lanes, cell_size=args.cell, **mode_kw)
    grid_kw = dict(rolling=0, min_cluster_points=20, raycast_wg_per_cu=args.raycast_wg, raycast_max_workgroups=args.raycast_max_wg, raycast_seg_items=args.raycast_seg or 0,
                   raycast_impl={"tiled": api.RAYCAST_TILED, "merge": api.RAYCAST_TILED_MERGE, "global": api.RAYCAST_GLOBAL}[args.raycast])
    grid = api.Grid(GRID, GRID, RES, **grid_kw)
    d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
    d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
    d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
    # the batch's initial poses arrive as one block [R (4 per scan) | t (2 per scan)]: one copy per step
    d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
    d_pose = api.DeviceArray(d_pose0.shape, np.float64)
    d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
    d_res = api.DeviceArray((S,), api.RESULT_DTYPE)

    # Pipelined: consecutive steps alternate over TWO grid buffers and two grid streams (every step ray-casts its batch into one
    # zeroed 2000 x 2000 grid of its own and finalizes it: steps are independent batches, as the count reset per step has always
    # made them), so that the grid update of step k+1 does not queue behind that of step k -- and, with N>1, runs while the rows of
    # step k are summed over the GPUs.  One grid (--one-grid): 0.352 ms per step; two: 0.342 (50 steps; 0.333 over 100).
    grids = [grid]
    if not args.no_pipeline and not args.one_grid:
        grids.append(api.Grid(GRID, GRID, RES, **grid_kw))

    # ---- how steps are launched
    # pipeline (default): CONSECUTIVE steps on three streams.  A0 / A1 in turn: initial poses in, registration of step k
    #   (touches no plane), two scans per workgroup (slam_icp_params::pair_scans: a batch of 256 then holds 128 CUs, and the
    #   batch after it starts on the other 128 instead of waiting for this one's slowest scan).  B: count reset, raycast,
    #   [N>1: exchange of the touched rows, RCCL sum of those rows,] finalize of step k-1, on the CUs the registrations
    #   leave.  One scan per workgroup on one stream: 0.53 ms per step; registration beside the grid update of the step
    #   before: 0.44; pairs on two registration streams: 0.36.  Every step still does all of its work; K steps are timed to
    #   completion.  With N>1 the host's wait for the 8-byte row range of step k-1 comes after step k's registration has
    #   been enqueued.
    # graph: one stream, one captured hipGraph replayed per step (N=1).  calls: one stream, call by call.
    launch = "pipeline" if not args.no_pipeline else ("calls" if (multi or args.no_graph or not args.warmup) else "graph")
    n_cu = api.device_info()[1]
    icp_one = icp                                       # library defaults: what the one-stream forms use
    if launch == "pipeline" and S < 2 * n_cu and args.lanes == 0:
        icp = api.Icp(m_ga, m_nga, max_iter=N_ITERS, min_delta=-1.0, cell_size=args.cell, pair_scans=2, **mode_kw)
    # three priority levels: never the same hardware queue (see mapper.hip)
    pq = args.private_queues
    calibrating = merging and args.backend == "nccl" and launch == "pipeline" and (world > 1 or args.calibrate) and not args.no_calibrate and \
        (args.reg_cu_cap is None or args.merge_order is None or args.grid_lag is None)
    cal_caps = [args.reg_cu_cap] if args.reg_cu_cap is not None else [0, 1, 2, 4]
    cal_orders = [args.merge_order] if args.merge_order is not None else ["late", "early"]
    cal_lags = [args.grid_lag] if args.grid_lag is not None else [3, 4, 6]
    if args.reg_cu_cap is None:
        args.reg_cu_cap = 0
    if args.merge_order is None:
        args.merge_order = "late"
    if args.grid_lag is None:
        args.grid_lag = 3
    max_lag = max(cal_lags) if calibrating else args.grid_lag

    def reg_streams(cap):
        return [api.Stream(priority=None if k % 2 == 0 else -1, reserve_cus_per_xcd=cap, private_queue=pq) for k in range(max(args.step_streams, 1))]
    SA = reg_streams(args.reg_cu_cap)
    sb = api.Stream(priority=1, private_queue=pq)
    SB = [sb] + [api.Stream(priority=1, private_queue=pq) for _ in grids[1:]]      # one grid stream per grid
    sa = SA[0]
    NB = max(4, args.step_streams + max_lag + 1)      # pose / result buffers: the registrations in flight plus the grid updates two steps behind
    pose = [d_pose] + [api.DeviceArray(d_pose0.shape, np.float64) for _ in range(NB - 1)]
    pR = [p_.view(0, batch.R.shape) for p_ in pose]
    pt = [p_.view(batch.R.size, batch.t.shape) for p_ in pose]
    res = [d_res] + [api.DeviceArray((S,), api.RESULT_DTYPE) for _ in range(NB - 1)]
    icp_done_own = [api.Event() for _ in range(NB)]
    icp_done = list(icp_done_own)
    grid_done = [api.Event() for _ in range(NB)]
    merge_rows_seen = []
    # (start, end) events around the registration launches of the timed region, on the stream of each
    live = [(api.Event(), api.Event()) for _ in range(min(args.steps, 64))]
    t_enq = [None] * len(live)           # host clock when the enqueue of registration k had returned (timed region)
    ev_ref = api.Event()                 # device clock <-> host clock: recorded on an idle stream just before the timed region

    d_R0, d_t0 = d_pose0.view(0, batch.R.shape), d_pose0.view(batch.R.size, batch.t.shape)

    def enqueue_icp(k, a, e=None, handle=None, timed=False):
        s_ = k % NB
        # the grid update NB steps ago has read these poses: normally long since -- asked on the host, so that no wait packet stands
        # in front of the launch then (20 steps 0.334 -> 0.332 ms per step, 100 steps 0.315 -> 0.314: tools/exp/query_ab.sh)
        if not grid_done[s_].query():
            a.wait_event(grid_done[s_])
        if timed and k < len(live):
            live[k][0].record(a)
        if e: e[0].record(a)
        # the batch's initial poses are read where they lie, the registered poses written to this step's buffer
        # (slam_icp_fit_batch_from_dev: no copy of the initial poses into the in/out arrays, no launch gap behind it)
        (handle or icp).fit_batch_from_dev(d_pts, d_off, d_nga, S, d_R0, d_t0, pR[s_], pt[s_], 5.0, res[s_], None, a)
        if e: e[1].record(a)
        # (a timed step's end-of-launch event is also the one its grid update waits for: one record behind the launch, not two)
        icp_done[s_] = live[k][1] if (timed and k < len(live)) else icp_done_own[s_]
        icp_done[s_].record(a)

    # N>1, pipelined: the merge of a step belongs to the communicator's helper thread (slam_grid_merge_async) -- the enqueue
    # thread posts it and goes on.  What stays on this thread is a host-to-host wait before a grid (and its stream) is used
    # again two steps later: by then the helper has long enqueued that merge (merge.merge_wait_ms says how long it was not so).
    use_thread = merging and bool(args.merge_thread)
    mode = {"thread": False}                            # set per run_steps call: only the pipelined, un-instrumented steps post to the helper
    ticket_of = {}                                      # id(grid) -> ticket of the merge that still owns the grid and its stream
    ray_done = {id(g_): api.Event() for g_ in grids}    # behind the raycast whose merge the ticket stands for
    pace = {"ms": 0.0, "n": 0}                          # the enqueue thread's wait for the DEVICE (back-pressure), apart from the merge's

    def settle(g):
        t_ = ticket_of.pop(id(g), None)
        if t_ is not None:
            # Two waits, told apart: (1) for the device to have finished the raycast this merge belongs to -- back-pressure: a bounded
            # pipeline whose consumer is the slower side holds its producer exactly here, by a whole step per step, and that is what
            # keeps the host a fixed number of registrations AHEAD (host_enqueue_slack_ms); (2) for the helper thread to have posted the
            # merge behind it -- the 24-byte MIN over the ranks, the helper's wake-up, the enqueue of the row all-reduce: the part that
            # grows with rank count and skew (merge.merge_wait_ms, from slam_comm_get_stats).
            t_p = time.perf_counter()
            ray_done[id(g)].synchronize()
            pace["ms"] += (time.perf_counter() - t_p) * 1e3
            pace["n"] += 1
            merge_rows_seen.append(comm.ticket_wait(t_))

    def enqueue_grid_update(k, b, e=None, g=None):
        """raycast of step k behind its registration [N>1: + the ranks' row ranges start travelling]; no host wait"""
        s_ = k % NB
        g = g or grid
        settle(g)
        b.wait_event(icp_done[s_])
        # (every step starts from zero counts: the step before on this grid ended with slam_grid_finalize_reset)
        g.raycast_scans_dev(d_pts, d_off, S, P, pR[s_], pt[s_], b)
        if e: e[2].record(b)
        if merging and not mode["thread"]:
            comm.merge_begin(g, b)

    def enqueue_grid_fold(k, b, e=None, g=None):
        """[N>1: the united row range of step k, the all-reduce of those rows,] finalize"""
        s_ = k % NB
        g = g or grid
        if mode["thread"]:
            # the poses of this step have been read once the raycast is through: that is all a later registration waits for
            grid_done[s_].record(b)
            ray_done[id(g)].record(b)
            ticket_of[id(g)] = comm.merge_async(g, b, api.MERGE_THEN_FINALIZE_RESET)
            return
        if merging:
            merge_rows_seen.append(comm.merge_finish(g, b))
        if e: e[3].record(b)
        g.finalize_reset(b)          # evidence + occupancy of this batch, count planes zero again: one launch
        if e: e[4].record(b)
        grid_done[s_].record(b)

    def enqueue_grid(k, b, e=None, g=None):
        enqueue_grid_update(k, b, e, g)
        enqueue_grid_fold(k, b, e, g)

    def run_steps(n, events=None, pipelined=True, handle=None, timed=False):
        if n <= 0:
            return
        E = (lambda k: events[k]) if events else (lambda k: None)
        mode["thread"] = use_thread and pipelined and not events
        # registrations the host stays ahead of the grid update it enqueues; with the helper thread "early" is simply no lag: the
        # step's raycast and its whole merge are posted with its registration (they wait for it on the device)
        lag = 0 if (mode["thread"] and args.merge_order == "early") else max(args.grid_lag, 1)
        if pipelined and merging and len(grids) == 2 and args.merge_order == "early" and not mode["thread"]:
            # N>1: a step's raycast and the start of its merge are enqueued WITH its registration (they wait for it on the
            # device); the host's wait for the united row range of step k-2 comes two registrations later, just before the
            # raycast of step k goes behind it on the same grid -- by then that range is normally back (merge.merge_wait_ms)
            for k in range(n):
                enqueue_icp(k, SA[k % len(SA)], E(k), handle, timed)
                if k >= 2:
                    enqueue_grid_fold(k - 2, SB[k % 2], E(k - 2), grids[k % 2])
                enqueue_grid_update(k, SB[k % 2], E(k), grids[k % 2])
            for k in range(max(n - 2, 0), n):
                enqueue_grid_fold(k, SB[k % 2], E(k), grids[k % 2])
        elif pipelined:
            # the host stays two registrations ahead of the grid update it enqueues
            for k in range(n):
                enqueue_icp(k, SA[k % len(SA)], E(k), handle, timed)
                if timed and k < len(t_enq):
                    t_enq[k] = time.perf_counter()
                if k == 0 and args.start_stagger_us > 0 and n > 1:
                    t_s = time.perf_counter()
                    while (time.perf_counter() - t_s) * 1e6 < args.start_stagger_us:
                        pass
                if k >= lag:
                    enqueue_grid(k - lag, SB[(k - lag) % len(SB)], E(k - lag), grids[(k - lag) % len(grids)])
            for k in range(max(n - lag, 0), n):
                enqueue_grid(k, SB[k % len(SB)], E(k), grids[k % len(grids)])
        else:
            for k in range(n):
                enqueue_icp(k, sa, E(k), handle, timed)
                enqueue_grid(k, sa, E(k), grids[k % len(grids)])

    for e_ in grid_done:
        e_.record(sb)
    for g_ in grids:
        g_.clear()
    calibration = None
    if calibrating:
        # which launch setting leaves RCCL's kernels room on THIS node: a few steps of each, the slowest rank's clock, the
        # fastest setting kept for the warm-up and the timed steps (every rank sees the same all-reduced times: same choice)
        calibration = {"tried": []}
        # (a registration stream that leaves CUs alone is refused -- SLAM_E_UNSUPPORTED -- on a device whose CU layout the mask was not
        # measured on, e.g. a partitioned one: such caps are dropped, on every rank alike, instead of ending the run in calibration)
        streams_of = {}
        for c in cal_caps:
            try:
                streams_of[c] = reg_streams(c)
            except api.SlamError as ex:
                if ex.code != api.E_UNSUPPORTED:
                    raise
        ok_caps = [1 if c in streams_of else 0 for c in cal_caps]
        if multi:
            ok_caps = [int(v) for v in rk.all_reduce(torch.tensor(ok_caps, dtype=torch.int64), torch.distributed.ReduceOp.MIN,
                                                     "registration-stream caps every rank supports").tolist()]
        calibration["caps_dropped_unsupported"] = [c for c, ok in zip(cal_caps, ok_caps) if not ok]
        cal_caps = [c for c, ok in zip(cal_caps, ok_caps) if ok]
        if not cal_caps:
            _RK.fail("no registration-stream setting is supported on this device")
        # late: the grid update (and the merge) of a step is enqueued `lag` registrations after its own; early: with it (no lag)
        settings = [(c, l, "late") for c in cal_caps for l in cal_lags if "late" in cal_orders] + \
                   [(c, 0, "early") for c in cal_caps if "early" in cal_orders]
        best_ms = {}
        run_steps(20)                  # the start-of-run transient (DESIGN Appendix A) belongs to no setting
        for rnd in range(3):           # interleaved rounds, the best round of a setting counts
            for cap_, lag_, order_ in settings:
                SA, args.merge_order = streams_of[cap_], order_
                args.grid_lag = lag_ or args.grid_lag
                sa = SA[0]
                run_steps(4)
                sync(); barrier()
                t_c = time.perf_counter()
                run_steps(30)
                sync(); barrier()
                best_ms[(cap_, lag_, order_)] = min(best_ms.get((cap_, lag_, order_), 1e9), rk.max_over_ranks(time.perf_counter() - t_c) / 30 * 1e3)
        calibration["tried"] = [{"reg_cu_cap_per_xcd": c, "grid_lag": l, "merge_order": o, "ms_per_step": best_ms[(c, l, o)]} for c, l, o in settings]
        # the first setting (the one-GPU default unless the command line says otherwise) stays unless another is 3 % faster
        best = min(calibration["tried"], key=lambda c_: c_["ms_per_step"])
        if best["ms_per_step"] > 0.97 * calibration["tried"][0]["ms_per_step"]:
            best = calibration["tried"][0]
        args.reg_cu_cap, args.merge_order = best["reg_cu_cap_per_xcd"], best["merge_order"]
        args.grid_lag = best["grid_lag"] or cal_lags[0]
        SA = streams_of[args.reg_cu_cap]
        sa = SA[0]
        calibration["kept"] = {"reg_cu_cap_per_xcd": args.reg_cu_cap, "grid_lag": args.grid_lag, "merge_order": args.merge_order}
        calibration["what"] = ("before the warm-up: three interleaved rounds of 30 steps per setting (registration streams leaving 0/1/2/4 CUs "
                               "per XCD alone x the grid update 3/4/6 registrations behind, or with its registration = early), the slowest "
                               "rank's clock, the best round of each; the first setting is kept unless another is 3 % faster; the timed "
                               "steps run with the one kept")
        merge_rows_seen.clear()
        sync()
        for g_ in grids:
            g_.clear()                 # (the update counter starts again: the warm-up's steps are counted below)
    run_steps(args.warmup, pipelined=launch == "pipeline")
    sync()
    upd_per_step = None
    if args.warmup:
        # every step starts from reset counts, but the update counter keeps running
        upd_per_step = sum(g_.total_updates() for g_ in grids) // args.warmup
    for g_ in grids:
        g_.clear()
    sync()
    barrier()
    sync()
    if comm is not None:
        comm.stats_reset()           # the merge statistics of the JSON line are those of the timed steps
    pace["ms"], pace["n"] = 0.0, 0
    graph = None
    if launch == "graph":
        try:
            graph = api.Graph(sa)
            with graph:      # the warm-up ran the same calls: every scratch buffer exists
                icp.fit_batch_from_dev(d_pts, d_off, d_nga, S, d_R0, d_t0, pR[0], pt[0], 5.0, d_res, None, sa)
                grid.raycast_scans_dev(d_pts, d_off, S, P, pR[0], pt[0], sa)
                grid.finalize(sa)            # (a captured graph replays fixed kernel arguments: finalize_reset alternates
                grid.reset_counts(sa)        # between two range buffers from call to call, so the graph keeps the two-call form)
            sync()
            graph.launch()   # one replay outside the timed region: a graph that cannot run must not cost the bench
            sync()
        except Exception as ex:   # fall back to launching call by call on a fresh stream
            print("hipGraph capture/replay failed (%s); launching call by call" % ex, file=sys.stderr)
            graph, launch = None, "calls"
            sa = sb = api.Stream()
            run_steps(1, pipelined=False)
            sync()
    if multi and args.test_kill_rank == rank:
        import signal
        os.kill(os.getpid(), signal.SIGKILL)
    import gc
    gc.collect()
    gc.disable()                      # (see run_config5: a full collection inside the timed region is an accident of the run)
    # The timed region -- exactly `steps` steps between a barrier + device synchronisation on either side -- is run `regions`
    # times back to back and the MEDIAN region is the one reported (ms_per_step, value); every region's figure goes into
    # ms_per_step_runs.  One region of 20 steps is 6 ms: a single sample of it carried the start-of-run transient of the first
    # launches (4-7 % between the driver's 20 steps and this file's 50); five regions cost 25 ms more and say how far they agree.
    region_s = []
    for region in range(max(args.regions, 1)):
        t_ref0 = time.perf_counter()
        ev_ref.record(SA[0] if launch == "pipeline" else sa)
        ev_ref.synchronize()
        t_ref1 = time.perf_counter()         # the event's device time lies between the two host readings
        t0 = time.perf_counter()
        if graph is not None:
            for k in range(args.steps):
                graph.launch()
        else:
            run_steps(args.steps, pipelined=launch == "pipeline", timed=True)
        sync()
        barrier()
        sync()
        region_s.append(time.perf_counter() - t0)
    if multi:
        region_s = [rk.max_over_ranks(e_) for e_ in region_s]     # every region by its slowest rank, then the median region
    elapsed = float(np.median(region_s))
    n_regions = len(region_s)
    gc.enable()
    pace_timed = dict(pace)
    for g_ in grids:
        settle(g_)
    merge_stats = comm.stats() if comm is not None else None
    if upd_per_step is None:      # no warm-up to count them in: the timed steps' own updates, before anything else runs
        upd_per_step = sum(g_.total_updates() for g_ in grids) // max(args.steps * n_regions, 1)
    live_ms = [a_.elapsed_ms(b_) for a_, b_ in live] if (graph is None and args.steps > 0) else []
    # How far ahead of the device the enqueue thread ran: registrations k and k + len(SA) share a stream, and that stream never runs
    # dry as long as the enqueue of the later one has returned before the earlier one ends on the device.  Device end times come
    # from the launches' own events, placed on the host's clock through ev_ref (read between two host readings just before t0).
    slack = None
    if live_ms and launch == "pipeline" and len(live) > len(SA):
        t_ref = 0.5 * (t_ref0 + t_ref1)
        end_host = [t_ref + ev_ref.elapsed_ms(b_) * 1e-3 for _, b_ in live]
        d_ = len(SA)
        sl_ = [(end_host[k - d_] - t_enq[k]) * 1e3 for k in range(d_, len(live)) if t_enq[k] is not None]
        if sl_:
            slack = {"min_ms": float(min(sl_)), "mean_ms": float(np.mean(sl_)), "registrations": len(sl_),
                     "clock_uncertainty_ms": (t_ref1 - t_ref0) * 0.5e3,
                     "what": "end of registration k - %d on the device minus the host time at which the enqueue of registration k (same "
                             "stream) had returned: positive = the launch was in the queue before its predecessor ended, the stream did not "
                             "run dry; over the first %d registrations of the timed region" % (d_, len(live))}
    # the same K steps one after the other on one stream (outside the timed region, N=1): what the pipelining buys
    seq_ms = None
    if launch == "pipeline" and not multi:
        run_steps(2, pipelined=False, handle=icp_one)
        sync()
        t1 = time.perf_counter()
        run_steps(args.steps, pipelined=False, handle=icp_one)
        sync()
        seq_ms = (time.perf_counter() - t1) / max(args.steps, 1) * 1e3
    # per-kernel times of the timed region's kernels, each alone on the chip: a few event-bracketed steps run one after
    # the other on one stream, outside the timed region
    ev = [[api.Event() for _ in range(5)] for _ in range(min(max(args.steps, 1), 10))]
    run_steps(len(ev), ev, pipelined=False)
    sync()
    d_R, d_t, d_res = pR[(len(ev) - 1) % NB], pt[(len(ev) - 1) % NB], res[(len(ev) - 1) % NB]
    grid = grids[(len(ev) - 1) % len(grids)]            # the grid of the last step: what the checks below read

    if multi:
        cnt = rk.all_reduce(torch.tensor([P, upd_per_step], dtype=torch.int64), what="sum of the ranks' points and updates")
        total_pts, total_upd = int(cnt[0].item()), int(cnt[1].item())
    if merging:
        # one more update, merged but not yet folded away (a step ends with finalize_reset, which zeroes the counts): the
        # merged planes hold every rank's updates of a step, once
        grid.raycast_scans_dev(d_pts, d_off, S, P, d_R, d_t, sb)
        comm.merge_begin(grid, sb)
        merge_rows_seen.append(comm.merge_finish(grid, sb))
        sync()
        hits, misses = grid.read_counts()
        merged = int(hits.astype(np.int64).sum() + misses.astype(np.int64).sum())
        assert merged == total_upd, "merged planes hold %d updates, the ranks made %d" % (merged, total_upd)
        grid.finalize_reset(sb)
        sync()
    if not multi:
        total_pts, total_upd = P, upd_per_step

    # per-kernel device time from the HIP events recorded on the launch stream
    seg = np.array([[e[i].elapsed_ms(e[i + 1]) for i in range(4)] for e in ev]) if len(ev) else np.zeros((1, 4))
    ms_icp, ms_ray, ms_merge, ms_fin = seg.mean(axis=0)

    # sanity on the result of the last step (not timed): all scans registered
    res = d_res.download()
    t_fin = d_t.download()
    assert (res["iters"] == N_ITERS).all(), "not every scan ran %d iterations" % N_ITERS
    pose_err = float(np.abs(t_fin - batch.true_poses[:, :2]).max())
    assert pose_err < 0.05 or args.map_kind != "room", "registered poses are off by %.3f m" % pose_err

    # the storage rows one step's finalize (and count reset) covers: the rows the step's raycast touched -- with N>1 the
    # rows any rank touched (the merge's united range); read back from the device-tracked range, outside the timed region
    if merging and merge_rows_seen:
        rows_lo, rows_hi = merge_rows_seen[-1]
    else:
        grid.raycast_scans_dev(d_pts, d_off, S, P, d_R, d_t, sa)
        sync()
        rows_lo, rows_hi = grid.dirty_rows()
        grid.finalize_reset(sa)
        sync()
    rows_cov = max(rows_hi - rows_lo + 1, 0)

    if rank == 0:
        info = icp.index_info()
        M = len(m_ga) + len(m_nga)
        # ALGORITHMIC bytes: ICP per scan 16*T + 8*M + 96, all iterations fused; the halo lists the
        # library builds for itself are NOT algorithmic bytes (reported beside, as index_bytes_read_per_launch)
        icp_bytes = 16 * P + S * (8 * M + 96)
        ray_bytes = 8 * upd_per_step + 16 * P           # 8 B RMW per cell update + 16 B per beam
        fin_bytes = rows_cov * GRID * 17                # per cell of the rows covered: 2x4 B counts in, 8 B evidence + 1 B occupancy out
        fused = bool(info.get("two_forms")) and args.lanes == 0
        paired = fused and (icp is not icp_one or S >= 2 * n_cu)
        icp_name = "icp_fit_pair_kernel" if paired else ("icp_fit_fused_kernel" if fused else "icp_fit_kernel")
        step_ms = elapsed / max(args.steps, 1) * 1e3
        # the registration launch as the timed region saw it (HIP events on its own stream around every launch; with two
        # registration streams two launches share the chip); ms_icp is the same kernel alone on the chip
        ms_icp_live = float(np.mean(live_ms)) if live_ms else float(ms_icp)
        ATOMIC_CEILING_GBS = 1300.0   # MI355X_MICROARCH.md, global atomics: ~1.3 TB/s of added bytes
        kernels = {
            icp_name: {"ms": ms_icp_live, "ms_alone_on_the_chip": float(ms_icp), "alg_bytes": icp_bytes,
                       "launches_in_flight": 2 if launch == "pipeline" else 1,
                       "index_bytes_read_per_launch": (S // 2 if paired else S) * (int(info.get("lds_bytes", 0)) + (int(info.get("list_bytes", 0)) if fused else 0))},
            "raycast_tiled_kernel (+ beams, work list)": {
                "ms": float(ms_ray), "alg_bytes": ray_bytes,
                "global_atomic_ceiling_GBps": ATOMIC_CEILING_GBS,
                "note": "alg_bytes = 8 B RMW per cell update + 16 B per beam; against the ceiling of one GLOBAL atomic per "
                        "update (SURVEY 8(d)): the updates are binned in LDS tiles and reach HBM as one atomic per touched "
                        "cell and segment, which is how the rate can exceed that ceiling"},
            "finalize_reset_rows_kernel": {"ms": float(ms_fin), "alg_bytes": fin_bytes, "rows_covered": rows_cov, "rows_of_the_grid": GRID,
                                     "note": "alg_bytes = 17 B x the cells of the storage rows the kernel covers (the touched rows, read "
                                             "back from the device-tracked range): counts in, evidence + occupancy out; the zeroes it "
                                             "writes back into the counts it has folded are not counted"},
        }
        prof_file, prof = pmc_profile(p2l)
        for k in kernels.values():
            k["GBps"] = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
            k["frac_of_hbm_peak"] = k["GBps"] / HBM_PEAK_GBS
        # the raycast's 8 B per cell update never reach HBM (the updates are binned in LDS): its algorithmic rate belongs beside
        # the global-atomic ceiling, and its HBM fraction comes from the MEASURED traffic of the committed counter passes
        kr = kernels["raycast_tiled_kernel (+ beams, work list)"]
        kr["alg_GBps_equivalent"] = kr.pop("GBps")
        kr["vs_global_atomic_ceiling"] = kr["alg_GBps_equivalent"] / ATOMIC_CEILING_GBS
        ray_traffic = sum(prof.get(n, {}).get("hbm_bytes_per_launch") or 0.0 for n in ("raycast_tiled_kernel", "beams_from_scans_kernel", "tile_items_wg_kernel"))
        kr["hbm_bytes_per_launch_measured"] = ray_traffic or None
        kr["GBps"] = (ray_traffic / (kr["ms"] * 1e-3) / 1e9) if (ray_traffic and kr["ms"] > 0 and (S, GRID) == (256, 2000)) else None
        kr["frac_of_hbm_peak"] = (kr["GBps"] / HBM_PEAK_GBS) if kr["GBps"] else None
        dom = max(kernels, key=lambda n: kernels[n]["ms"])
        pk = prof.get(dom.split(" ")[0], {})
        busy, lanes = pk.get("valu_busy_frac"), pk.get("valu_active_lane_share")
        held = min(n_cu, (S + 1) // 2 if paired else S) if dom == icp_name else n_cu
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS,
                "traffic": pk.get("hbm_bytes_per_launch"),
                "traffic_source": ("%s (rocprofv3 --pmc passes of this command, committed; not measured in this run; the "
                                   "profiler runs one dispatch at a time, so these are the kernel alone on the chip)" % prof_file)
                if prof_file else None,
                "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "avg_launch_ms": kernels[dom]["ms"],
                "alg_bytes_formula": "16*P + S*(8*M + 96) (SURVEY 8(d)); P=%d points, S=%d scans, M=%d model points" % (P, S, M),
                "per_step": {"GBps": kernels[dom]["alg_bytes"] / (step_ms * 1e-3) / 1e9,
                             "frac": kernels[dom]["alg_bytes"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "what": "alg_bytes_per_launch / ms_per_step: one launch of this kernel per step, whatever overlaps it"},
                # the bound this kernel actually runs against: the share of the chip's VALU lane-slots that do work
                "valu": {"bound": "valu", "frac": (busy * lanes) if (busy is not None and lanes is not None) else None,
                         "busy_frac": busy, "active_lane_share": lanes,
                         "cus_held_by_one_launch": held,
                         "frac_on_held_cus": (busy * lanes * n_cu / held) if (busy is not None and lanes is not None) else None,
                         "source": prof_file,
                         "profile_csrc_sha256": prof.get("_meta", {}).get("csrc_sha256"),
                         "run_csrc_sha256": csrc_sha(),
                         "stale": (prof.get("_meta", {}).get("csrc_sha256") != csrc_sha()) if prof_file else None,
                         "stale_means": "the committed counter summary was taken from kernel sources other than the ones this run was built "
                                        "from (sha256 over slam_amd/csrc): the fractions describe that build",
                         "meaning": "busy_frac = SIMD cycles (all 1024 SIMDs) that issued a VALU instruction; active_lane_share = "
                                    "SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); frac = their product = the share of the "
                                    "chip's VALU lane-slots that did work during a launch"},
                "note": "frac = alg_bytes_per_launch / avg_launch_ms / peak, avg_launch_ms = HIP events around every launch of the "
                        "timed region on its own stream (profiles/: the kernel's AverageNs in the kernel-trace summary of the same "
                        "command).  The kernel is VALU-issue bound (exact 1-NN search in LDS, DESIGN.md 4.1), not HBM-bound: "
                        "`valu` is the bound it runs against; the HBM fraction is reported because the contract asks for it."}
        out = {
            "metric": "registered_scan_points_per_s", "value": total_pts * args.steps / elapsed,
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True,
            "ms_per_step_runs": [e_ / max(args.steps, 1) * 1e3 for e_ in region_s],
            "timed_regions": {"n": n_regions, "reported": "median",
                              "what": "the timed region (`steps` steps between barrier + device synchronisation) run n times back to back behind "
                                      "the one warm-up; ms_per_step and value are the MEDIAN region's, ms_per_step_runs lists all of them in order"},
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 pose / f32 distance / int32 counts",
            "data": "synthetic",
            "icp_step": "point-to-line, 3x3 normal equations per iteration (icpPointToPlane.cpp:37-107; SLAM_ICP_P2L)" if p2l else
                        "point-to-point, 2x2 closed form (icpPointToPoint.cpp:33-172; SLAM_ICP_P2P: what the reference compiles)",
            "launch": {"pipeline": "consecutive steps on %d streams: registrations (two scans per workgroup) alternate on two, the grid "
                                   "update of the step before runs on %s" % (2 + len(SB), "the third" if len(SB) == 1 else
                                                                            "two more in turn, each step into a zeroed grid buffer of its own (two buffers)"),
                       "graph": "hipGraph replay of one captured step", "calls": "one stream, call by call"}[launch],
            "config": {"workload": "BASELINE config %s per GPU: %d x 1081-beam scans (%d points), %d ICP "
                                   "iterations vs %d-point map, Bresenham raycast into %dx%d @%.2f m, "
                                   "finalize%s" % ("2" if (S, GRID) == (256, 2000) else ("4" if (S, GRID) == (1024, 4000) else "2 (resized)"),
                                                   S, P, N_ITERS, M, GRID, GRID, RES,
                                                   ", all-reduce of the touched rows of the int32 planes (%s)" %
                                                   ("slam_grid_merge_begin/_finish over RCCL" if args.backend == "nccl" else "slam_grid_merge_begin/_finish over the host-staged communicator, gloo rehearsal") if merging
                                                   else (", NO merge (--no-merge: independent ranks)" if multi else "")),
                       "scans_per_gpu": S, "icp_iters": N_ITERS, "grid": [GRID, GRID], "grid_buffers": len(grids), "resolution": RES,
                       "map_points": M, "map_kind": args.map_kind, "map_points_per_class": [len(m_ga), len(m_nga)],
                       "icp_index": info, "raycast": args.raycast,
                       "raycast_worklist": grid.raycast_stats(),
                       "merge_rows": list(merge_rows_seen[-1]) if merging and merge_rows_seen else None,
                       "reg_cu_cap_per_xcd": args.reg_cu_cap, "grid_lag": args.grid_lag, "merge_order": args.merge_order if merging else None,
                       "merge_thread": bool(use_thread) if merging else None,
                       "calibration_skipped": bool(args.no_calibrate) if merging else None},
            "grid_cell_updates_per_s": total_upd * args.steps / elapsed,
            "cell_updates_per_step": total_upd,
            "point_iterations_per_s": total_pts * N_ITERS * args.steps / elapsed,
            "one_stream": None if seq_ms is None else
            {"ms_per_step": seq_ms, "value": total_pts / (seq_ms * 1e-3),
             "what": "the same steps call by call on ONE stream, one scan per workgroup, nothing overlapped (measured after "
                     "the timed region)"},
            "kernel_ms": {"icp": float(ms_icp), "raycast": float(ms_ray), "merge": float(ms_merge),
                          "finalize": float(ms_fin)},
            "kernels": kernels,
            "host_enqueue_slack_ms": slack,
            "roofline": roof,
            "max_pose_error_m": pose_err,
            "device": api.device_info()[0],
        }
        if p2l and world == 1:
            # the point-to-line handle's own costs: its model build (index + the normals of icpPointToPlane.cpp:279-305, one
            # 10-neighbour scatter matrix per model point, + a normal per halo-list entry) and one scan through the host API
            out["model_build_ms"] = model_build_times(api, synth, m_ga, m_nga, **mode_kw)
            out["single_scan"] = single_scan_times(api, synth, m_ga, m_nga, **mode_kw)
        if multi:
            # what a scaling curve is read with: how many ranks the transport itself counts, what the exchange step moved and
            # what it cost -- per timed step, on rank 0 (the ranks move the same rows: the all-reduce is over their union)
            ms_ = merge_stats or {}
            n_m = max(ms_.get("merges", 0), 1)
            out["rccl_ranks"] = ms_.get("n_ranks") if merging else None
            out["merge"] = None if not merging else {
                "transport": "rccl" if ms_.get("transport") == 0 else "host-staged (gloo rehearsal)",
                "rccl_version": ms_.get("rccl_version"), "ranks": ms_.get("n_ranks"),
                "merges_in_timed_region": ms_.get("merges"), "timed_regions": n_regions,
                "rows_per_merge": ms_.get("rows", 0) / n_m, "bytes_per_merge_per_rank": ms_.get("bytes", 0) / n_m,
                "merge_wait_ms": ms_.get("wait_ms", 0.0) / n_m,
                "backpressure_wait_ms": (pace_timed["ms"] / max(pace_timed["n"], 1)) if use_thread else None,
                "helper_wait_ms": ms_.get("helper_wait_ms", 0.0) / n_m,
                "merges_by_helper_thread": ms_.get("async_merges"),
                "rccl_channels": rccl_channels() if args.backend == "nccl" else None,
                "allreduce_ms": (ms_.get("allreduce_ms", 0.0) / ms_["timed"]) if ms_.get("timed") else None,
                "allreduce_GBps_per_rank": (ms_.get("bytes", 0) / n_m) / (ms_["allreduce_ms"] / ms_["timed"] * 1e-3) / 1e9
                if ms_.get("timed") and ms_.get("allreduce_ms") else None,
                "what": "merge_wait_ms = time the ENQUEUE thread spent waiting for anything of a merge (per merge; with the helper "
                        "thread: for a ticket or a place in its queue, AFTER the device had finished the raycast the merge belongs to; "
                        "without: inside slam_grid_merge_finish for the united row range, the raycast's device time included); "
                        "backpressure_wait_ms = the enqueue thread's wait, just before that, for the device to finish that raycast (two "
                        "steps old: what holds a producer that is faster than the chip; host_enqueue_slack_ms says how far ahead it stays); "
                        "helper_wait_ms = the helper thread's wait for the united row range (device time of raycast + 24-byte all-reduce, hidden "
                        "from the enqueue thread); rccl_channels = workgroups per RCCL collective kernel, from RCCL's INIT log; "
                        "allreduce_ms = HIP events around the grouped row all-reduces on the grid stream (includes the time their "
                        "kernels waited for a CU beside the registration workgroups); both from slam_comm_get_stats"}
            if merging:
                out["merge"]["calibration"] = calibration
            out["no_merge"] = bool(args.no_merge)
        if world == 1 and not args.no_extras:
            # the model build reported separately; metric (1) with the transfers in; the reference's own
            # usage (one scan per match); config 3's match
            out["model_build_ms"] = model_build_times(api, synth, m_ga, m_nga)
            dt, Rs, ts, st = stream_rate(api, synth, m_ga, m_nga, batch, GRID)
            assert np.abs(ts - batch.true_poses[:, :2]).max() < 0.05
            out["value_pcie_inclusive"] = P / dt
            out["pcie_inclusive"] = {"ms_per_chunk": dt * 1e3, "chunk_scans": S,
                                     "path": "slam_mapper_push/_wait: pinned host chunk -> H2D | ICP | raycast on three "
                                             "streams -> poses D2H; 48 chunks, three in flight, filling and draining included"}
            out["single_scan"] = single_scan_times(api, synth, m_ga, m_nga)
            try:
                out["endpoint"] = endpoint_leg(api, synth, batch, d_R.download().reshape(-1, 4), t_fin, GRID, RES, not args.no_cpu_baseline)
            except Exception as ex:   # the headline line must not depend on the extra leg
                out["endpoint"] = {"error": repr(ex)}
            try:
                c3 = run_config3(args.clouds, not args.no_cpu_baseline)
                out["config3"] = {k: c3[k] for k in ("workload", "target", "mean_xy_error_m", "max_xy_error_m", "ms_per_cloud_chain", "cpp_adapter", "ms_per_cloud",
                                                     "clouds_per_s", "stepwise_clouds_per_s", "model_points", "mean_icp_iterations", "target_model_ms")}
            except Exception as ex:   # the headline line must not depend on the extra leg
                out["config3"] = {"error": str(ex)}
            try:   # the grid half one cloud at a time through the C++ drop-in, as local_mapper runs it
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                import mls_time
                out["local_mapper_cloud"] = mls_time.measure(4)
            except Exception as ex:
                out["local_mapper_cloud"] = {"error": str(ex)}
            # short runs of the other GPU configs of BASELINE.json, each in a process of its own (this one stays idle):
            # one GPU's share of config 4 (1024 scans into 4000 x 4000, pipelined steps) and config 5 (the streaming mapper
            # with its sliding-window target, PCIe inclusive)
            # the solver north_star names (point-to-line, 3x3 normal equations) on the same workload, same pipelined step, the CPU
            # oracle's point-to-line step timed beside it
            out["p2l"] = side_run(["--mode", "p2l", "--steps", "30", "--warmup", "5", "--no-extras"],
                                  ("value", "ms_per_step", "icp_step", "grid_cell_updates_per_s", "kernel_ms", "roofline", "one_stream", "model_build_ms", "single_scan",
                                   "max_pose_error_m", "cpu_baseline"))
            out["config4_share"] = side_run(["--config", "4", "--steps", "12", "--warmup", "3", "--no-extras", "--no-cpu-baseline"],
                                            ("value", "ms_per_step", "grid_cell_updates_per_s", "cell_updates_per_step", "kernel_ms", "max_pose_error_m"))
            # (the whole 10 240-scan stream of BASELINE config 5: a leg of 4096 scans is 16 chunks, a quarter of them filling and
            # draining the pipeline -- 0.43 ms per chunk where the stream runs at 0.41)
            out["config5"] = side_run(["--config", "5", "--stream-scans", "10240"],
                                      ("value", "ms_per_step", "steps", "grid_cell_updates_per_s", "max_pose_error_m"))
            # the N > 1 path with two ranks on THIS GPU: bench.py launching its own ranks, the library's merge entry points over
            # its host-staged communicator (gloo carries the host buffers; RCCL does not put two ranks on one device).  Not a
            # scaling number -- two ranks share one chip and stage through the host -- but the same code path as --gpus N.
            out["two_ranks_on_one_gpu"] = side_run(["--gpus", "2", "--backend", "gloo", "--one-device", "--steps", "6", "--warmup", "2",
                                                    "--no-cpu-baseline"], ("n_gpus", "value", "ms_per_step", "cell_updates_per_step"))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(m_ga, m_nga, batch, GRID, RES, p2l)
        emit(out)
    if multi:
        rk.barrier("end of run")
        if comm is not None:
            comm.close()
        rk.close()


def run_config5(args, api, synth, m_ga, m_nga, rank, world, comm, rk, sync, barrier):
    """BASELINE config 5: a stream of scans per GPU through slam_mapper_* -- pinned chunks -> H2D | ICP | raycast on
    three streams, sliding-window target (last 4 chunks, rebuilt every 4), merge over the GPUs + finalize every 8
    chunks.  PCIe inclusive by construction."""
    chunk, n_total = args.chunk, args.stream_scans
    n_chunks = max(1, n_total // chunk)
    # every rank streams its own stretch of ONE loop of n_chunks * chunk * world scans: every chunk is new ground
    # (a repeated chunk would fill the sliding window with exact duplicates: distance ties, the slow exact pass)
    log("rank %d: generating %d chunks of %d scans" % (rank, n_chunks, chunk))
    chunks = [synth.make_batch(chunk, n_loop=n_chunks * chunk * world, first=(rank * n_chunks + k) * chunk) for k in range(n_chunks)]
    base = chunks[0]
    # the local map stays small enough for the index to live in LDS (about 10 k points): a 5 k-point prior map plus
    # at most 5 k points of the last four chunks
    m_ga, m_nga = synth.make_map(5000)
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=0, min_cluster_points=20, raycast_wg_per_cu=args.raycast_wg), grid_size_x=GRID, grid_size_y=GRID,
                    resolution=RES, max_scans=chunk, max_points=max(c.n_points for c in chunks),
                    icp=dict(max_iter=N_ITERS, min_delta=-1.0, list_min_halo=args.list_min_halo, pair_scans=args.pair_scans,
                             **(dict(mode=api.ICP_P2L, normals_k=10) if args.mode == "p2l" else {})),
                    window_chunks=args.window, rebuild_every=args.rebuild_every, keep_prior=1, target_points=5000,
                    thin_res=args.thin, merge_every=args.merge_every, registration_streams=args.reg_streams, slots=args.slots)
    if comm is not None:
        mp.use_comm(comm)
    # warm-up: enough chunks for one whole rebuild cycle (the first rebuild creates its stream, loads the build's kernels and
    # sizes the pool: 2-3 ms once per mapper, a sixth of a 16-chunk leg if it is left inside the timed region)
    n_warm = max(mp.n_slots, min(args.rebuild_every + 2, 10) if args.window else 0)
    n_warm = -(-n_warm // mp.n_slots) * mp.n_slots
    for w0 in range(0, n_warm, mp.n_slots):
        for s in [mp.push(base) for _ in range(mp.n_slots)]:
            mp.wait(s)
    sync()
    upd_warm = mp.grid.total_updates()       # the warm-up chunks' updates are not the timed chunks'
    barrier()
    # (no cyclic garbage collection inside the timed region: with torch imported a full collection is 30 ms, and where the interpreter's
    # allocation count puts it is an accident of the run -- chunks of 512 scans had it between the last push and the first wait, every
    # time: 2.3 ms per chunk instead of 0.57, tools/exp/c5_torch_gap.py)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    pending, worst = [], 0.0
    for k in range(n_chunks):
        if len(pending) == mp.n_slots:
            slot, j = pending.pop(0)
            R, t = mp.wait(slot)
            worst = max(worst, float(np.abs(t - chunks[j].true_poses[:, :2]).max()))
        pending.append((mp.push(chunks[k]), k))
    for slot, j in pending:
        R, t = mp.wait(slot)
        worst = max(worst, float(np.abs(t - chunks[j].true_poses[:, :2]).max()))
    mp.finish()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    upd = mp.grid.total_updates() - upd_warm
    st = mp.stats()
    if rk is not None:
        elapsed = rk.max_over_ranks(elapsed)
    if rank == 0:
        pts = sum(c.n_points for c in chunks) * world
        emit({
            "metric": "registered_scan_points_per_s", "value": pts / elapsed, "unit": "points/s", "n_gpus": world,
            "steps": n_chunks, "warmup": n_warm, "ms_per_step": elapsed / n_chunks * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 pose / f32 distance / int32 counts", "data": "synthetic",
            "icp_step": "point-to-line (SLAM_ICP_P2L)" if args.mode == "p2l" else "point-to-point (SLAM_ICP_P2P)",
            "config": {"workload": "BASELINE config 5 per GPU: %d scans streamed in chunks of %d from pinned host memory, %d ICP "
                                   "iterations against %s, Bresenham raycast into %dx%d @%.2f m, merge over the GPUs + finalize every %d chunks"
                                   % (n_chunks * chunk, chunk, N_ITERS,
                                      ("a sliding-window target (5 k-point prior map + at most 5 k points of the last %d chunks thinned at %.2f m, "
                                       "rebuilt on the device every %d)" % (args.window, args.thin, args.rebuild_every)) if args.window
                                      else "the fixed 5 k-point prior map",
                                      GRID, GRID, RES, args.merge_every),
                       "pcie_inclusive": True, "mapper": st, "target_index": mp.target_index_info()},
            "grid_cell_updates_per_s": upd * world / elapsed, "max_pose_error_m": worst, "device": api.device_info()[0]})
    mp.close()
    if rk is not None:
        rk.barrier("end of run")
        if comm is not None:
            comm.close()
        rk.close()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as ex:      # with N > 1: say which rank failed and why, tell the others, exit non-zero -- never hang
        if _RK is not None:
            import traceback
            traceback.print_exc()
            _RK.fail("%s: %s" % (type(ex).__name__, ex))
        raise
