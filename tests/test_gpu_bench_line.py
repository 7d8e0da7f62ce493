"""The one JSON line of `python bench.py` at N = 1: the fields the round's driver reads, and the arithmetic a reader can
redo from the line itself -- value from ms_per_step, roofline.frac = achieved / peak with achieved = algorithmic bytes of one
launch / its measured duration, no per-kernel rate above the HBM peak."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract_and_arithmetic():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-extras",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must be ONE JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "registered_scan_points_per_s" and d["unit"] == "points/s" and d["n_gpus"] == 1
    assert d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    pts = 276242                                               # config 2's 256 scans
    assert abs(d["value"] - pts / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the figure is the MEDIAN of five timed regions of `steps` steps each, all of them in the line
    runs = d["ms_per_step_runs"]
    assert len(runs) == 5 == d["timed_regions"]["n"] and abs(sorted(runs)[2] - d["ms_per_step"]) < 1e-9
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    k = d["kernels"]["icp_fit_pair_kernel"]
    assert abs(r["achieved"] - k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]      # one launch, its own duration
    assert k["alg_bytes"] == 16 * pts + 256 * (8 * 10000 + 96)                                     # SURVEY 8(d), per launch
    for name, kk in d["kernels"].items():
        assert (kk.get("GBps") or 0.0) <= 8000.0, (name, kk)
    assert d["max_pose_error_m"] < 0.05


def _bench(*argv):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must be ONE JSON line: %r" % (lines[:3],)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_force_dist_runs_the_n_gt_1_pipeline_over_a_real_rccl_communicator():
    """The N > 1 code path -- RCCL communicator made from an ncclUniqueId, slam_grid_merge_begin / _finish in every pipelined
    step -- with the one rank a one-GPU box has, and the fields a scaling curve is read with."""
    d = _bench("--force-dist", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["no_merge"] is False
    m = d["merge"]
    assert m["transport"] == "rccl" and m["rccl_version"] > 20000 and m["ranks"] == 1
    assert m["timed_regions"] == 5 and m["merges_in_timed_region"] == 4 * 5        # five timed regions of `steps` steps each
    lo, hi = d["config"]["merge_rows"]
    assert 0 <= lo <= hi < 2000 and abs(m["rows_per_merge"] - (hi - lo + 1)) < 8       # the room's rows, step after step
    assert m["bytes_per_merge_per_rank"] == m["rows_per_merge"] * 2000 * 8            # [hits | misses] int32 of those rows
    assert m["merge_wait_ms"] >= 0.0 and m["allreduce_ms"] > 0.0
    assert d["max_pose_error_m"] < 0.05
    # round 5: the merges are the helper thread's (slam_grid_merge_async); the enqueue thread does not wait for the device
    assert d["config"]["merge_thread"] is True and m["merges_by_helper_thread"] >= 4
    # (what the enqueue thread waits for is told apart: the device finishing a raycast two steps old -- back-pressure, a step per step
    # once the producer is ahead -- and, after that, the helper thread posting the merge behind it: the part ranks and skew add to)
    # (the merge's own part, with one rank: the key kernel, a 24-byte all-reduce, the helper's wake-up and its enqueue -- 0.05-0.1 ms, under a step)
    assert m["merge_wait_ms"] < 0.25 and m["helper_wait_ms"] > 0.0 and m["backpressure_wait_ms"] >= 0.0
    sl = d["host_enqueue_slack_ms"]
    assert sl is not None and sl["registrations"] >= 1 and sl["min_ms"] <= sl["mean_ms"]
    # ... and the same steps with the merge on the enqueue thread (rounds 2-4), still there behind a switch
    d0 = _bench("--force-dist", "--merge-thread", "0", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline")
    assert d0["config"]["merge_thread"] is False and d0["merge"]["merges_by_helper_thread"] == 0
    assert d0["merge"]["merges_in_timed_region"] == 4 * 5 and d0["config"]["merge_rows"] == d["config"]["merge_rows"]


@pytest.mark.gpu
def test_bench_calibrates_the_n_gt_1_launch_setting():
    """With more than one rank the bench first times reg-cu-cap 0/1/2/4 x (grid-lag 3/4/6 | early) and keeps the fastest (no
    multi-GPU lease has said which one RCCL's kernels need); --calibrate runs that with the one rank of this box."""
    d = _bench("--force-dist", "--calibrate", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline")
    c = d["merge"]["calibration"]
    assert [(t["reg_cu_cap_per_xcd"], t["grid_lag"], t["merge_order"]) for t in c["tried"]] == \
        [(cap, lag, "late") for cap in (0, 1, 2, 4) for lag in (3, 4, 6)] + [(cap, 0, "early") for cap in (0, 1, 2, 4)]
    best = min(c["tried"], key=lambda t: t["ms_per_step"])
    if best["ms_per_step"] > 0.97 * c["tried"][0]["ms_per_step"]:       # the default stays unless another setting is 3 % faster
        best = c["tried"][0]
    assert c["kept"] == {"reg_cu_cap_per_xcd": best["reg_cu_cap_per_xcd"], "grid_lag": best["grid_lag"] or 3, "merge_order": best["merge_order"]}
    assert d["config"]["reg_cu_cap_per_xcd"] == c["kept"]["reg_cu_cap_per_xcd"] and d["config"]["merge_order"] == c["kept"]["merge_order"]
    assert all(0.2 < t["ms_per_step"] < 5.0 for t in c["tried"])
    assert d["merge"]["merges_in_timed_region"] == 4 * 5 and d["max_pose_error_m"] < 0.05 and d["merge"]["calibration"]["caps_dropped_unsupported"] == []
    # a setting given on the command line is not calibrated over
    d = _bench("--force-dist", "--calibrate", "--merge-order", "late", "--grid-lag", "3", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline")
    assert [(t["reg_cu_cap_per_xcd"], t["grid_lag"], t["merge_order"]) for t in d["merge"]["calibration"]["tried"]] == \
        [(0, 3, "late"), (1, 3, "late"), (2, 3, "late"), (4, 3, "late")]


@pytest.mark.gpu
def test_bench_no_merge_and_reserved_cus():
    """--no-merge: the same ranks with the exchange step left out (what the merge costs is the difference to the default
    run); --reg-cu-cap: registration streams that leave CUs of every XCD to the other streams' kernels."""
    d = _bench("--force-dist", "--no-merge", "--reg-cu-cap", "1", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline")
    assert d["no_merge"] is True and d["merge"] is None and d["rccl_ranks"] is None
    assert d["config"]["reg_cu_cap_per_xcd"] == 1 and "NO merge" in d["config"]["workload"]
    assert d["max_pose_error_m"] < 0.05 and d["value"] > 1e8
