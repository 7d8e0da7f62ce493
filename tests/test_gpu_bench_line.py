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
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    k = d["kernels"]["icp_fit_pair_kernel"]
    assert abs(r["achieved"] - k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]      # one launch, its own duration
    assert k["alg_bytes"] == 16 * pts + 256 * (8 * 10000 + 96)                                     # SURVEY 8(d), per launch
    for name, kk in d["kernels"].items():
        assert kk.get("GBps", 0.0) <= 8000.0, (name, kk)
    assert d["max_pose_error_m"] < 0.05
