#!/usr/bin/env python3
"""the batch registration kernels against MODEL SIZE -- 10 k (BASELINE configs 1/2/4), 20 k and the reference's
own cap of 2 x 19 999 points (icpTools.h:21, icpTools.cpp:255-274, icp.cpp:51-60) -- on config 2's 256 scans x 30 iterations.
Per model: the pipelined step (pairs, two launches in flight) and the registration launch alone on the chip (pair and fused
form), point-iterations/s, and whether the index lives in LDS or is read from HBM/L2.  One child `bench.py` per line; writes
gpurun_out/model_size.json (copied to profiles/rNN_model_size.json)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "model_size.json")


def bench(*argv):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": p.stderr[-600:]}
    return json.loads(lines[-1])


def main():
    steps = sys.argv[1] if len(sys.argv) > 1 else "40"
    rows = []
    for kind, m in (("room", 10000), ("room", 20000), ("room", 39998), ("uniform", 10000), ("uniform", 39998)):
        common = ["--map-points", str(m), "--map-kind", kind, "--steps", steps, "--warmup", "5", "--no-extras", "--no-cpu-baseline"]
        d = bench(*common)                                  # pipelined: pairs, two registration launches in flight
        f = bench(*common, "--no-pipeline")                 # one stream: the fused form, one scan per workgroup, alone on the chip
        row = {"map_kind": kind, "map_points": m}
        if "error" in d or "error" in f:
            row["error"] = d.get("error") or f.get("error")
        else:
            pts = 276242
            row.update({
                "points_per_class": d["config"]["map_points_per_class"],
                "index": {k: d["config"]["icp_index"].get(k) for k in ("in_lds", "lds_bytes", "list_bytes", "cell", "nx", "ny", "two_forms")},
                "pipelined_ms_per_step": d["ms_per_step"], "pipelined_point_iterations_per_s": d["point_iterations_per_s"],
                "pair_launch_ms_in_the_mix": d["roofline"]["avg_launch_ms"], "pair_launch_ms_alone": d["kernel_ms"]["icp"],
                "pair_point_iterations_per_s_alone": pts * 30 / (d["kernel_ms"]["icp"] * 1e-3),
                "fused_launch_ms_alone": f["kernel_ms"]["icp"],
                "fused_point_iterations_per_s_alone": pts * 30 / (f["kernel_ms"]["icp"] * 1e-3),
                "one_stream_ms_per_step": f["ms_per_step"],
                "max_pose_error_m": d["max_pose_error_m"],
            })
        rows.append(row)
        print(json.dumps(row), flush=True)
    base = next((r for r in rows if r.get("map_kind") == "room" and r.get("map_points") == 10000 and "error" not in r), None)
    for r in rows:
        if base and "error" not in r:
            r["pair_alone_vs_10k_room"] = r["pair_launch_ms_alone"] / base["pair_launch_ms_alone"]
            r["pipelined_vs_10k_room"] = r["pipelined_ms_per_step"] / base["pipelined_ms_per_step"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump({"what": __doc__, "steps": int(steps), "rows": rows}, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
