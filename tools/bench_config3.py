#!/usr/bin/env python3
"""BASELINE config 3 end to end: 50 clouds of 64 x 2048 rays along the loop through the CCICP chain, two drivers:
the Python-driven device-resident chain registers every cloud against the FIRST one (fixed target: errors grow with the
distance along the loop); the C++ adapter (tests/cpp/ccicp_sequence.cpp) replaces the target by the cloud just matched
every 10 clouds, as scan_registration does when graph_slam publishes a new map.  Both say which in their JSON
(`target`).  The chain: (ground segmentation, GA/NGA classification, voxel filter, crop + split, class-
constrained ICP with the reference defaults max_iter 20 / min_delta 1e-6, height recovery), device-resident
through the C-ABI (host loop in Python: test/bench plumbing).  Prints one JSON line (not the headline bench)."""
import ctypes as C
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import api, synth


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measure_cpp(clouds, poses, advance=10, passes=3, forms=("seq",)):
    """The same sequence through the C++ drop-in (include/slam_amd/ccicp.hpp, what ros/scan_registration_node.cpp calls where
    the reference calls icpTools.cpp:222-298): tests/cpp/ccicp_sequence.cpp compiled with g++ against the shipped library,
    clouds handed over as files, the target advanced every `advance` clouds.  Returns the program's JSON line as a dict."""
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "slam_amd", "lib")
    # (children of a profiled run must not inherit the profiler: g++ and the C++ program would run with its library preloaded
    # and leave kernel-stats files of their own beside the parent's)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTX"))}
    # (... nor bench.py's GPU_MAX_HW_QUEUES=8, which is for ITS streams: the program stands for a user's, with the runtime's defaults --
    # CCICP::matchSequence deals its scene chains over four streams for the runtime's four queues: 0.23-0.25 ms per match there, 0.30-0.32 with 8)
    env.pop("GPU_MAX_HW_QUEUES", None)
    keep = os.environ.get("SLAM_C3_KEEP_DIR")      # tools/profile_round.sh: the binary and its inputs stay there, for rocprofv3 on the binary itself
    import contextlib
    if keep:
        os.makedirs(keep, exist_ok=True)
    with (contextlib.nullcontext(keep) if keep else tempfile.TemporaryDirectory()) as d:
        exe = os.path.join(d, "ccicp_sequence")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "ccicp_sequence.cpp"), "-o", exe,
                               "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], env=env)
        init, truth = [], []
        for k in range(1, len(clouds)):
            j = ((k - 1) // advance) * advance if advance > 0 else 0        # the cloud that is the target when k is matched
            pa, pb = poses[j], poses[k]
            ca, sa = np.cos(pa[2]), np.sin(pa[2])
            rel = (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])
            yaw = rel[2] + 0.02
            init.append([rel[0] + 0.1, rel[1] - 0.1, 0.0, 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
            truth.append(list(rel))
        for k, c in enumerate(clouds):
            np.ascontiguousarray(c, np.float32).tofile(os.path.join(d, "cloud%d.f32" % k))
        np.array(init, np.float64).tofile(os.path.join(d, "init.f64"))
        np.array(truth, np.float64).tofile(os.path.join(d, "truth.f64"))
        out = None
        for form in forms:
            p = subprocess.run([exe, d, str(len(clouds)), str(advance), str(passes), form], capture_output=True, text=True, timeout=600, env=env)
            if p.returncode != 0:
                raise RuntimeError("ccicp_sequence %s failed (%d): %s" % (form, p.returncode, p.stderr[-500:]))
            line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            got = np.fromfile(os.path.join(d, "poses_out.f64"), np.float64).reshape(-1, 7)
            if out is None:
                out = line
                out["poses"] = got
            else:
                # a throughput form beside the sequential one: its rate, and that it gives the same poses
                form = {"seqp": "seq_pinned_clouds"}.get(form, form)
                out.setdefault("throughput_forms", {})[form] = {
                    "ms_per_match": line["ms_per_match"], "clouds_per_s": line["clouds_per_s"],
                    "ms_per_cloud_with_target_updates": line["ms_per_cloud_with_target_updates"],
                    "max_xy_error_m": line["max_xy_error_m"],
                    "max_pose_difference_vs_seq": {"xy_m": float(np.abs(got[:, :2] - out["poses"][:, :2]).max()),
                                                   "z_m": float(np.abs(got[:, 2] - out["poses"][:, 2]).max()),
                                                   "quat": float(np.abs(got[:, 3:] - out["poses"][:, 3:]).max())}}
                f = out["throughput_forms"][form]
                f["same_poses_as_seq"] = bool(max(f["max_pose_difference_vs_seq"].values()) < 1e-9)
                if not f["same_poses_as_seq"]:
                    print("bench_config3: form %s DIFFERS from the sequential form: %r" % (form, f["max_pose_difference_vs_seq"]), file=sys.stderr)
        out["truth"] = np.array(truth)
        out["init"] = np.array(init)
        out["target_of"] = [((k - 1) // advance) * advance if advance > 0 else 0 for k in range(1, len(clouds))]
        return out


def measure(n_clouds=50, cell=0.0, dump_case=None, advance=10):
    """Runs the sequence twice (the first pass warms buffers and code objects) and returns the summary dict."""
    L = api.lib()
    seg, cc = api.GroundSegmentation(), api.Ccicp()
    clouds, poses = zip(*[synth.make_cloud3d(k, n_loop=50) for k in range(n_clouds)])
    n_max = max(len(c) for c in clouds)
    d_xyz = api.DeviceArray((n_max, 3), np.float32)
    d_lab = api.DeviceArray((n_max,), np.uint8)
    d_obs = api.DeviceArray((n_max, 4), np.float32)
    d_gnd = api.DeviceArray((n_max, 4), np.float32)
    d_flag = api.DeviceArray((n_max,), np.uint8)
    d_cloud = api.DeviceArray((n_max, 4), np.float32)
    d_ga = api.DeviceArray((20000, 2), np.float64)
    d_nga = api.DeviceArray((20000, 2), np.float64)
    h_ga = np.zeros((20000, 2)); h_nga = np.zeros((20000, 2))

    def front_end(xyz, voxel, pose_xy):
        """segmentGround + classifyPoints (+ voxel filter) + crop/split -> (ga, nga) host arrays, ground count."""
        n = len(xyz)
        api.check(L.slam_memcpy_h2d(d_xyz.ptr, xyz.ctypes.data, xyz.nbytes, None))
        seg.segment_dev(d_xyz, n, 3, d_lab)
        n_obs, n_gnd, n_out = C.c_int(0), C.c_int(0), C.c_int(0)
        api.check(L.slam_ccicp_select_dev(cc.h, d_xyz.ptr, n, 3, d_lab.ptr, (1 << 2) | (1 << 3), d_obs.ptr, C.byref(n_obs), None))
        api.check(L.slam_ccicp_select_dev(cc.h, d_xyz.ptr, n, 3, d_lab.ptr, 1 << 1, d_gnd.ptr, C.byref(n_gnd), None))
        api.check(L.slam_gseg_classify_ga_dev(seg.h, d_obs.ptr, n_obs.value, 4, d_flag.ptr, None))
        if voxel:
            api.check(L.slam_ccicp_voxel_downsample_dev(cc.h, d_obs.ptr, d_flag.ptr, n_obs.value, 4, 0.5, 0.5, 2.0, d_cloud.ptr,
                                                        n_max, C.byref(n_out), None))
        else:
            api.check(L.slam_ccicp_bin_order_dev(cc.h, d_obs.ptr, d_flag.ptr, n_obs.value, 4, d_cloud.ptr, C.byref(n_out), None))
        counts = (C.c_int * 2)()
        crop = pose_xy is not None
        api.check(L.slam_ccicp_split_dev(cc.h, d_cloud.ptr, n_out.value, 4, 1 if crop else 0, pose_xy[0] if crop else 0.0,
                                         pose_xy[1] if crop else 0.0, 75.0, 20000, d_ga.ptr, d_nga.ptr, counts, None))
        api.check(L.slam_memcpy_d2h(h_ga.ctypes.data, d_ga.ptr, 16 * counts[0], None))
        api.check(L.slam_memcpy_d2h(h_nga.ctypes.data, d_nga.ptr, 16 * counts[1], None))
        return h_ga[:counts[0]].copy(), h_nga[:counts[1]].copy(), n_gnd.value

    def relative(pa, pb):
        ca, sa = np.cos(pa[2]), np.sin(pa[2])
        return (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])

    def run():
        t0 = time.perf_counter()
        m_ga, m_nga, n_gnd_t = front_end(clouds[0], False, (0.0, 0.0))       # setTargetCloud (SCAN_TO_SCAN) + crop
        t1 = time.perf_counter()
        icp = api.Icp(m_ga, m_nga, cell_size=cell)                           # max_iter 20, min_delta 1e-6 (icp.cpp:27)
        t_create = time.perf_counter() - t1
        t_model = time.perf_counter() - t0
        # ground_target stays in d_gnd only until the next front_end: keep a copy for the height recovery
        d_gt = api.DeviceArray((max(n_gnd_t, 1), 4), np.float32)
        api.check(L.slam_memcpy_d2d(d_gt.ptr, d_gnd.ptr, 16 * n_gnd_t, None))
        errs, iters, t_front, t_icp, t_h, n_scene = [], [], 0.0, 0.0, 0.0, 0
        for k in range(1, n_clouds):
            rel = relative(poses[0], poses[k])
            a = time.perf_counter()
            s_ga, s_nga, _ = front_end(clouds[k], True, None)                # setSceneCloud
            b = time.perf_counter()
            R0, t0_ = synth.pose_to_Rt(rel[0] + 0.1, rel[1] - 0.1, rel[2] + 0.02)
            if k == 1 and dump_case:                                         # the arrays of one match, for offline analysis
                np.savez(dump_case, m_ga=m_ga, m_nga=m_nga, s_ga=s_ga, s_nga=s_nga, R0=R0, t0=t0_)
            R, t, res = icp.fit(s_ga, s_nga, R0, t0_)                        # doICPMatch: IcpPointToPoint::fit
            c = time.perf_counter()
            yaw = np.arctan2(R[1, 0], R[0, 0])
            pose7 = (C.c_double * 7)(t[0], t[1], 0.0, 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2))
            z = C.c_double(0.0)
            api.check(L.slam_ccicp_height_dev(cc.h, d_gt.ptr, n_gnd_t, 4, pose7, C.byref(z), None, None, None))
            d = time.perf_counter()
            t_front += b - a; t_icp += c - b; t_h += d - c
            n_scene += len(s_ga) + len(s_nga)
            errs.append(np.hypot(t[0] - rel[0], t[1] - rel[1])); iters.append(res.iters)
        icp.close()
        return t_model, t_create, t_front, t_icp, t_h, errs, iters, len(m_ga) + len(m_nga), n_scene

    def run_chain():
        """the same sequence through the device-resident chain: per cloud one H2D of the cloud, slam_ccicp_scene_dev ->
        slam_icp_fit_batch_dev (the cloud's size never leaves the device) -> slam_ccicp_height_pose_dev, one read-back"""
        m_ga, m_nga, n_gnd_t = front_end(clouds[0], False, (0.0, 0.0))
        icp = api.Icp(m_ga, m_nga, cell_size=cell)
        d_gt = api.DeviceArray((max(n_gnd_t, 1), 4), np.float32)
        api.check(L.slam_memcpy_d2d(d_gt.ptr, d_gnd.ptr, 16 * n_gnd_t, None))
        d_ngt = api.DeviceArray.from_host(np.array([n_gnd_t], np.int32))
        d_pts = api.DeviceArray((2 * 20000, 2), np.float64)
        d_scan, d_counts = api.DeviceArray((3,), np.int32), api.DeviceArray((4,), np.int32)
        d_pose = api.DeviceArray((6,), np.float64)
        d_R, d_t = d_pose.view(0, (1, 4)), d_pose.view(4, (1, 2))
        d_res = api.DeviceArray((1,), api.RESULT_DTYPE)
        d_z = api.DeviceArray((2,), np.float64)
        h_pose = np.zeros(6)
        st = api.Stream()
        ev = [(api.Event(), api.Event()) for _ in range(n_clouds)]
        errs, iters, t_all = [], [], 0.0
        for k in range(1, n_clouds):
            rel = relative(poses[0], poses[k])
            R0, t0_ = synth.pose_to_Rt(rel[0] + 0.1, rel[1] - 0.1, rel[2] + 0.02)
            xyz = clouds[k]
            a = time.perf_counter()
            h_pose[:4], h_pose[4:] = R0.reshape(4), t0_
            api.check(L.slam_memcpy_h2d(d_xyz.ptr, xyz.ctypes.data, xyz.nbytes, None))
            api.check(L.slam_memcpy_h2d_async(d_pose.ptr, h_pose.ctypes.data, 48, st.ptr))
            ev[k][0].record(st)
            api.check(L.slam_ccicp_scene_dev(cc.h, seg.h, d_xyz.ptr, len(xyz), 3, 1, 0, 0.0, 0.0, 75.0, 20000, d_pts.ptr, d_scan.ptr,
                                             None, d_counts.ptr, st.ptr))
            ev[k][1].record(st)
            icp.fit_batch_dev(d_pts, d_scan, d_scan.view(2, (1,)), 1, d_R, d_t, 5.0, d_res, None, st)
            api.check(L.slam_ccicp_height_pose_dev(cc.h, d_gt.ptr, d_ngt.ptr, n_gnd_t, 4, d_R.ptr, d_t.ptr, 0.0, d_z.ptr, st.ptr))
            st.synchronize()
            pose, res, z = d_pose.download(), d_res.download()[0], d_z.download()
            t_all += time.perf_counter() - a
            errs.append(np.hypot(pose[4] - rel[0], pose[5] - rel[1])); iters.append(int(res["iters"]))
            chain_poses.append((pose[4], pose[5], np.arctan2(pose[2], pose[0]), z[0], int(res["iters"]), int(res["n_corr"])))
        icp.close()
        chain_front_ms[:] = [ev[k][0].elapsed_ms(ev[k][1]) for k in range(1, n_clouds)]
        return t_all, errs, iters

    chain_front_ms = []                                                      # device time of slam_ccicp_scene_dev per cloud (HIP events)
    chain_poses = []                                                         # (x, y, yaw, z, iterations, correspondences) per match
    run()                                                                    # warm-up: buffers, code objects
    t_model, t_create, t_front, t_icp, t_h, errs, iters, n_model, n_scene = run()
    run_chain()
    del chain_poses[:]
    t_chain, errs_c, iters_c = run_chain()
    assert iters_c == iters, "the chain and the stepwise path ran different iteration counts"
    assert np.abs(np.array(errs_c) - np.array(errs)).max() < 1e-9
    n = n_clouds - 1
    total = t_front + t_icp + t_h
    try:
        # first with the target fixed (what the Python-driven chain above did): the adapter must hand back the same poses
        same = measure_cpp(clouds[:min(n_clouds, 6)], poses[:min(n_clouds, 6)], 0, passes=1)
        want = np.array(chain_poses[:len(same["poses"])])
        got = same["poses"]
        yaw = 2.0 * np.arctan2(got[:, 5], got[:, 6])
        dyaw = np.abs((yaw - want[:, 2] + np.pi) % (2 * np.pi) - np.pi)
        assert np.abs(got[:, :2] - want[:, :2]).max() < 1e-9 and dyaw.max() < 1e-9 and np.abs(got[:, 2] - want[:, 3]).max() < 1e-9, \
            "the C++ adapter and the Python-driven chain disagree"
        cpp = measure_cpp(clouds, poses, advance, forms=("seq", "seqp", "ahead", "batch"))
        if "throughput_forms" in cpp:
            cpp["throughput_forms"]["what"] = (
                "beside the sequential form (one cloud at a time from pageable memory: the reference's usage and this adapter's default): 'seq_pinned_clouds' = the same calls on clouds in pinned memory (slam_host_alloc: the upload only enqueues); 'ahead' = "
                "CCICP::prepareSceneCloud(cloud k+1) before doICPMatch(cloud k), the next cloud's upload and scene chain on a second "
                "stream; 'batch' = CCICP::matchSequence, the clouds between two target replacements with their scene chains on four "
                "streams and their fits as ONE slam_icp_fit_batch_dev (initial poses known beforehand); clouds in pinned host memory")
        cpp_detail = {k: cpp.pop(k) for k in ("poses", "truth", "init", "target_of")}
        cpp["target"] = "replaced by the cloud just matched every %d clouds (setTargetCloud): a match is against a cloud at most %d poses back" % (advance, advance)
        cpp["equals_python_chain_on_fixed_target"] = True
        cpp["what"] = ("the same clouds through the C++ drop-in slam_amd::CCICP (include/slam_amd/ccicp.hpp; tests/cpp/ccicp_sequence.cpp "
                       "compiled against the shipped library): setSceneCloud + doICPMatch per cloud, the target replaced by the cloud "
                       "just matched every %d clouds (setTargetCloud, SCAN_TO_SCAN); wall clock per match incl. the cloud's H2D" % advance)
    except Exception as ex:     # no g++ on the box, ...: the Python-driven chain above still stands
        cpp, cpp_detail = {"error": repr(ex)}, None
    return {
        "_cpp_detail": cpp_detail,          # poses / truth / initial poses / target of every match of the C++ leg (bench.py: the oracle beside it)
        "target": "fixed: every cloud is registered against cloud 0 (up to %d poses away along the loop)" % n,
        "mean_xy_error_m": float(np.mean(errs_c)), "max_xy_error_m": float(np.max(errs_c)),
        "metric": "registered_clouds_per_s", "value": n / t_chain, "unit": "clouds/s", "steps": n, "warmup": n,
        "ms_per_step": t_chain / n * 1e3,
        "ms_per_cloud_chain": round(t_chain / n * 1e3, 3),
        "cpp_adapter": cpp,
        "chain": "slam_ccicp_scene_dev -> slam_icp_fit_batch_dev -> slam_ccicp_height_pose_dev on one stream: one H2D of the cloud "
                 "(pageable host memory), one read-back of pose / result / height per cloud; identical results to the stepwise path",
        "stepwise_clouds_per_s": n / total,
        "config": {"workload": "BASELINE config 3: %d clouds x %d rays registered against the first through the CCICP chain "
                               "(ground segmentation, GA/NGA classification, voxel filter, crop + split, class-constrained "
                               "ICP max_iter 20 / min_delta 1e-6, height recovery), one match at a time, device-resident chain; ms_per_cloud = the same "
                               "through the stepwise host API"
                               % (n, len(clouds[0]))},
        "workload": "config 3: %d clouds x %d rays registered against the first through the CCICP chain" % (n, len(clouds[0])),
        "ms_per_cloud": {"front end, device chain (slam_ccicp_scene_dev: 9 launches, HIP events)": round(float(np.mean(chain_front_ms)), 3),
                         "front end (segment, classify, voxel, split)": round(t_front / n * 1e3, 3),
                         "icp fit (one scan, host API)": round(t_icp / n * 1e3, 3), "height": round(t_h / n * 1e3, 3),
                         "total": round(total / n * 1e3, 3)},
        "clouds_per_s": n / t_chain, "rays_per_s": n * len(clouds[0]) / t_chain,
        "scene_points_per_match": n_scene / n, "registered_scene_points_per_s": n_scene / t_icp,
        "target_model_ms": round(t_model * 1e3, 3), "target_index_build_ms": round(t_create * 1e3, 3), "model_points": n_model,
        "mean_icp_iterations": float(np.mean(iters)),
        "median_xy_error_m_within_5_clouds": float(np.median(errs[:5]))}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    cell = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # model lattice pitch (0 = library default)
    out = measure(n, cell, os.environ.get("SLAM_DUMP_CASE"))
    out.pop("_cpp_detail", None)             # numpy arrays: for bench.py's oracle leg, not for the JSON line
    print(json.dumps(out))
