#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on MI355X.

One "step" = one pass of the per-scan hot path over one batch resident in HBM
(BASELINE config 2 per GPU): 256 synthetic 1081-beam scans, each registered
against the 10k-point map with 30 fixed ICP iterations (min_delta = -1), then
ray-cast from their registered poses into one 2000 x 2000 @ 0.05 m grid
(Bresenham free space + hits), then the counts are folded into the evidence /
occupancy planes (finalize).  With N > 1 every rank does that for its own 256
scans (weak scaling) and the int32 hit/miss planes are merged with one RCCL
all-reduce per step before finalize.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` is registered scan-points/s over the whole
job; `grid_cell_updates_per_s` is the second half of BASELINE.json's metric.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SCANS, N_ITERS, GRID, RES, MAP_POINTS = 256, 30, 2000, 0.05, 10000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(m_ga, m_nga, batch, grid_size, res):
    """Times the CPU oracle (a port of the reference path; tests pin it) on this
    host: ICP over a bounded sample of the same scans with OpenMP over scans,
    plus the Bresenham update of those scans.  Test infrastructure used as a
    reported baseline only -- never on the measured path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    n_s = min(batch.n_scans, 128)
    sub = batch.shard(0, max(1, batch.n_scans // n_s)) if n_s < batch.n_scans else batch
    model = O.IcpModel(m_ga, m_nga)
    p = O.icp_params(N_ITERS, -1.0, 5.0, O.NN_KDTREE)
    # repeated until about 20 core-seconds have gone into it, so that thread start-up and the clock do not matter
    reps, t_icp = 0, 0.0
    while t_icp * threads < 20.0 and reps < 200:
        t0 = time.perf_counter()
        R, t, iters, ncorr, delta = model.fit_batch(sub.pts, sub.scan_off, sub.scan_nga, sub.R, sub.t, p,
                                                    n_threads=threads)
        t_icp += time.perf_counter() - t0
        reps += 1
    t_icp /= reps
    # single-thread rate on a smaller sample (the reference's own execution model)
    one = sub.shard(0, max(1, sub.n_scans // 16))
    t0 = time.perf_counter()
    model.fit_batch(one.pts, one.scan_off, one.scan_nga, one.R, one.t, p, n_threads=1)
    t_icp1 = time.perf_counter() - t0
    g = O.grid_params(grid_size, grid_size, res, min_cluster_points=20)
    hits = np.zeros(grid_size * grid_size, np.int32)
    misses = np.zeros(grid_size * grid_size, np.int32)
    ends, origins = [], []
    for s in range(sub.n_scans):
        o, e = sub.scan_off[s], sub.scan_off[s + 1]
        ends.append(O.transform_points(sub.pts[o:e], R[s], t[s]))
        origins.append(np.tile(t[s].astype(np.float32), (e - o, 1)))
    ends, origins = np.concatenate(ends), np.concatenate(origins)
    t0 = time.perf_counter()
    _, _, upd = O.grid_raycast(g, origins, ends, hits, misses, n_threads=threads)
    t_grid = time.perf_counter() - t0
    t0 = time.perf_counter()
    n1 = len(ends) // 8
    _, _, upd1 = O.grid_raycast(g, origins[:n1], ends[:n1], hits, misses)
    t_grid1 = time.perf_counter() - t0
    # contended atomics can make the threaded grid update slower than one thread: the baseline takes the faster
    t_grid_best = min(t_grid, t_grid1 * upd / max(upd1, 1))
    return {
        "value": sub.n_points / (t_icp + t_grid_best), "unit": "points/s", "cores": threads,
        "kind": "port",
        "sample": "%d x (%d of the %d scans x %d ICP iterations; kd-tree NN, OpenMP over scans, %d threads) "
                  "+ Bresenham of the same scans into the %dx%d grid (the faster of OpenMP over beams with "
                  "atomic increments and one thread)"
                  % (reps, sub.n_scans, batch.n_scans, N_ITERS, threads, grid_size, grid_size),
        "icp_points_per_s": sub.n_points / t_icp,
        "icp_points_per_s_1thread": one.n_points / t_icp1,
        "grid_cell_updates_per_s": upd / t_grid,
        "grid_cell_updates_per_s_1thread": upd1 / t_grid1,
        "host_cores": cores,
    }


def pmc_traffic(kernel, field="hbm_bytes_per_launch"):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same
    command (profiles/rNN_traffic.json, written by tools/summarize_profiles.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); None if not profiled.
    field "valu_busy_frac": share of SIMD cycles that issued a VALU instruction (same file)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        t = json.load(open(files[-1]))
        key = kernel if kernel in t else ("raycast_tiled_kernel" if kernel.startswith("raycast") else None)
        return t[key].get(field) if key else None
    except Exception:
        return None


def main():
    global GRID
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scans", type=int, default=N_SCANS, help="scans per GPU (default: config 2; config 4: 1024)")
    ap.add_argument("--grid", type=int, default=GRID, help="grid side in cells (default: config 2; config 4: 4000)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per scan point (0 = library default)")
    ap.add_argument("--cell", type=float, default=0.0, help="ICP cell pitch in metres (0 = library default)")
    ap.add_argument("--raycast", choices=["tiled", "global"], default="tiled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="N=1: launch every step call by call instead of replaying one captured hipGraph")
    ap.add_argument("--no-torch", action="store_true", help="N=1 only: do not import torch at all")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the N>1 path "
                         "with several ranks on ONE GPU: --one-device)")
    ap.add_argument("--one-device", action="store_true", help="all ranks use GPU 0 (rehearsal with --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the N>1 code path (process group + all-reduce) with one rank")
    args = ap.parse_args()
    GRID = args.grid

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a launch through torch.distributed.run" % args.gpus)
        args.gpus = world

    torch = dist = None
    if world > 1 or not args.no_torch:
        # torch first, so that this process runs ONE HIP runtime (torch's) for
        # both the library's kernels and RCCL
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
    from slam_amd import api, synth
    api.set_device(local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    # ---- synthetic inputs of BASELINE config 2 (per rank: its own 256 scans of the loop)
    S = args.scans
    m_ga, m_nga = synth.make_map(MAP_POINTS)
    batch = synth.make_batch(S, n_loop=S * world, first=rank * S)
    P = batch.n_points
    icp = api.Icp(m_ga, m_nga, max_iter=N_ITERS, min_delta=-1.0, lanes_per_point=args.lanes, cell_size=args.cell)
    grid = api.Grid(GRID, GRID, RES, rolling=0, min_cluster_points=20,
                    raycast_impl=api.RAYCAST_TILED if args.raycast == "tiled" else api.RAYCAST_GLOBAL)
    d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
    d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
    d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
    # the batch's initial poses arrive as one block [R (4 per scan) | t (2 per scan)]: one copy per step
    d_pose0 = api.DeviceArray.from_host(np.concatenate([batch.R.ravel(), batch.t.ravel()]), np.float64)
    d_pose = api.DeviceArray(d_pose0.shape, np.float64)
    d_R, d_t = d_pose.view(0, batch.R.shape), d_pose.view(batch.R.size, batch.t.shape)
    d_res = api.DeviceArray((S,), api.RESULT_DTYPE)

    planes = None
    if multi:
        # zero-copy view of the library's [hits | misses] planes for the collective
        ptr, n_ints = grid.counts_dev()

        class _Planes:
            __cuda_array_interface__ = {"shape": (n_ints,), "typestr": "<i4", "data": (ptr, False),
                                        "version": 2, "strides": None}
        planes = torch.as_tensor(_Planes(), device=torch.device("cuda", local_rank))
        assert planes.data_ptr() == ptr

    def sync():
        api.synchronize()
        if torch is not None:
            torch.cuda.synchronize()

    def barrier():
        if multi:
            dist.barrier()

    ev = [[api.Event() for _ in range(5)] for _ in range(args.steps)]

    # N=1: all calls go to one created stream, so that a step can be captured into a hipGraph and replayed
    # with a single launch (about nine launches per step otherwise).
    # N>1: two streams.  A: poses in, ICP, then -- once the previous step's finalize has released the planes --
    # count reset and raycast.  B: the RCCL sum of the planes and finalize.  The registration of step k+1 (which
    # does not touch the planes) runs while step k's planes are merged over xGMI and finalized.
    st = None if multi else api.Stream()
    if multi:
        s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
        a, b = s_a.cuda_stream, s_b.cuda_stream
        ev_ray, ev_fin = api.Event(), api.Event()      # planes written by the raycast / released by finalize
        L = api.lib()
    else:
        a = b = st

    def step(e=None):
        # one batch: initial poses in, a fresh local count map, register, ray-cast, merge over the GPUs, finalize
        d_pose.copy_from(d_pose0, a)
        if e: e[0].record(a)
        icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, a)
        if e: e[1].record(a)
        if multi:
            api.check(L.slam_stream_wait_event(a, ev_fin.ptr))   # no-op before the first finalize
        grid.reset_counts(a)
        grid.raycast_scans_dev(d_pts, d_off, S, P, d_R, d_t, a)
        if e: e[2].record(a)
        if multi:
            ev_ray.record(a)
            api.check(L.slam_stream_wait_event(b, ev_ray.ptr))
            with torch.cuda.stream(s_b):
                for part in merge_parts:  # RCCL sum of the int32 planes over xGMI (the touched rows of both planes)
                    dist.all_reduce(part)
        if e: e[3].record(b)
        grid.finalize(b)
        if e: e[4].record(b)
        if multi:
            ev_fin.record(b)

    merge_parts = [planes] if multi else []
    grid.clear()
    for _ in range(args.warmup):
        step()
    sync()
    merge_rows = None
    if multi and args.warmup:
        # The planes are row-major: the rows any rank touched are one contiguous slice of each plane.  The
        # warm-up steps (full-plane merges) show which rows that is; the timed steps merge those rows only
        # (room of 30 m in a 100 m grid: about a third of the bytes).  Checked after the timed loop: nothing
        # outside the slice, and the merged total equals the ranks' updates.
        both = planes.view(2, GRID, GRID)
        rows = (both != 0).any(dim=2).any(dim=0).nonzero()
        y0 = int(rows.min().item()) if rows.numel() else 0
        y1 = int(rows.max().item()) + 1 if rows.numel() else 0
        yr = torch.tensor([-y0, y1], dtype=torch.int64, device="cuda")
        dist.all_reduce(yr, op=dist.ReduceOp.MAX)
        y0, y1 = max(0, -int(yr[0].item()) - 16), min(GRID, int(yr[1].item()) + 16)
        merge_rows = (y0, y1)
        merge_parts = [both[0, y0:y1].reshape(-1), both[1, y0:y1].reshape(-1)]
        assert all(p_.is_contiguous() and p_.data_ptr() == planes.data_ptr() + 4 * (k_ * GRID * GRID + y0 * GRID)
                   for k_, p_ in enumerate(merge_parts))
    upd_per_step = None
    if args.warmup:
        upd_per_step = grid.total_updates() // args.warmup
    grid.clear()
    sync()
    barrier()
    sync()
    graph = None
    if st is not None and not args.no_graph and args.warmup:
        try:
            graph = api.Graph(st)
            with graph:      # the warm-up ran the same calls: every scratch buffer exists
                step()
            sync()
            graph.launch()   # one replay outside the timed region: a graph that cannot run must not cost the bench
            sync()
        except Exception as ex:   # fall back to launching call by call on a fresh stream
            print("hipGraph capture/replay failed (%s); launching call by call" % ex, file=sys.stderr)
            graph = None
            st = api.Stream()
            step()
            sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if graph is not None:
            graph.launch()
        else:
            step(ev[k])
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if graph is not None:    # per-kernel times from a few event-bracketed steps outside the timed region
        ev = ev[:min(len(ev), 10)]
        for e in ev:
            step(e)
        sync()

    if upd_per_step is None:
        upd_per_step = grid.total_updates() // max(args.steps, 1)
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([P, upd_per_step], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt)
        total_pts, total_upd = int(cnt[0].item()), int(cnt[1].item())
        # the merged planes of the last step hold every rank's updates of that step, once
        merged = int(planes.to(torch.int64).sum().item())
        assert merged == total_upd, "merged planes hold %d updates, the ranks made %d" % (merged, total_upd)
        if merge_rows is not None:
            inside = int(sum(p_.to(torch.int64).sum().item() for p_ in merge_parts))
            assert inside == merged, "updates outside the merged rows %s" % (merge_rows,)
    else:
        total_pts, total_upd = P, upd_per_step

    # per-kernel device time from the HIP events recorded on the launch stream
    seg = np.array([[e[i].elapsed_ms(e[i + 1]) for i in range(4)] for e in ev]) if args.steps else np.zeros((1, 4))
    ms_icp, ms_ray, ms_merge, ms_fin = seg.mean(axis=0)

    # sanity on the result of the last step (not timed): all scans registered
    res = d_res.download()
    t_fin = d_t.download()
    assert (res["iters"] == N_ITERS).all(), "not every scan ran %d iterations" % N_ITERS
    pose_err = float(np.abs(t_fin - batch.true_poses[:, :2]).max())
    assert pose_err < 0.05, "registered poses are off by %.3f m" % pose_err

    if rank == 0:
        info = icp.index_info()
        M = len(m_ga) + len(m_nga)
        # algorithmic bytes: per scan 16*T + 8*M + 96, all iterations fused
        icp_bytes = 16 * P + S * (8 * M + 96)
        ray_bytes = 8 * upd_per_step + 16 * P           # 8 B RMW per cell update + 16 B per beam
        fin_bytes = GRID * GRID * 17                     # 2x4 B counts in, 8 B evidence + 1 B occupancy out
        kernels = {}
        fused = bool(info.get("two_forms")) and args.lanes == 0
        if fused:
            # one launch reads the scans, the cell index (8 B/point + starts), the halo lists and the poses
            kernels["icp_fit_fused_kernel (ring search, then list sweeps)"] = {
                "ms": float(ms_icp), "alg_bytes": icp_bytes + S * int(info.get("list_bytes", 0))}
        else:
            kernels["icp_fit_kernel"] = {"ms": float(ms_icp), "alg_bytes": icp_bytes}
        kernels.update({
            "raycast_tiled_kernel (+ beams, work list)": {"ms": float(ms_ray), "alg_bytes": ray_bytes},
            "finalize_kernel": {"ms": float(ms_fin), "alg_bytes": fin_bytes},
        })
        for k in kernels.values():
            k["GBps"] = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        dom = max(kernels, key=lambda n: kernels[n]["ms"])
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom.split(" ")[0]),
                "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "avg_launch_ms": kernels[dom]["ms"],
                "valu_busy_frac": pmc_traffic(dom.split(" ")[0], "valu_busy_frac"),
                "note": "the ICP kernel is VALU-issue bound (exact 1-NN search in LDS, one workgroup per scan, "
                        "launch time = slowest scan), not HBM-bound: valu_busy_frac (PMC) is the share of SIMD "
                        "cycles that issued a VALU instruction; see DESIGN.md 4.1"}
        out = {
            "metric": "registered_scan_points_per_s", "value": total_pts * args.steps / elapsed,
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 pose / f32 distance / int32 counts",
            "data": "synthetic", "launch": "hipGraph replay of one captured step" if graph is not None else "call by call",
            "config": {"workload": "BASELINE config %s per GPU: %d x 1081-beam scans (%d points), %d ICP "
                                   "iterations vs %d-point map, Bresenham raycast into %dx%d @%.2f m, "
                                   "finalize%s" % ("2" if (S, GRID) == (256, 2000) else ("4" if (S, GRID) == (1024, 4000) else "2 (resized)"),
                                                   S, P, N_ITERS, M, GRID, GRID, RES,
                                                   ", %s all-reduce of the touched rows of the int32 planes" % ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)") if multi else ""),
                       "scans_per_gpu": S, "icp_iters": N_ITERS, "grid": [GRID, GRID], "resolution": RES,
                       "map_points": M, "icp_index": info, "raycast": args.raycast,
                       "raycast_worklist": grid.raycast_stats(),
                       "merge_rows": list(merge_rows) if multi and merge_rows else None},
            "grid_cell_updates_per_s": total_upd * args.steps / elapsed,
            "cell_updates_per_step": total_upd,
            "point_iterations_per_s": total_pts * N_ITERS * args.steps / elapsed,
            "kernel_ms": {"icp": float(ms_icp), "raycast": float(ms_ray), "merge": float(ms_merge),
                          "finalize": float(ms_fin)},
            "kernels": kernels,
            "roofline": roof,
            "max_pose_error_m": pose_err,
            "device": api.device_info()[0],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(m_ga, m_nga, batch, GRID, RES)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
