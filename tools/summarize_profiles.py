#!/usr/bin/env python3
"""Turns gpurun_out/profile_<tag>/ (written by tools/profile_round.sh on the GPU
box) into the committed summaries under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of bench.py
  <tag>_pmc.csv            mean per-dispatch counter values per kernel
  <tag>_traffic.json       HBM bytes per launch per kernel from FETCH_SIZE / WRITE_SIZE
                           (MI355X_MICROARCH.md: units of KiB; FETCH_SIZE counts 64 B per
                           128-B request on wide coalesced reads -> read side doubled)
  <tag>_bench.json         the un-profiled bench line of the same build
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profile_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def kname(k):
    m = re.search(r"(\w+_kernel)", k)
    if not m:
        return k.split("(")[0][:48]
    name = m.group(1)
    if name == "icp_fit_kernel":  # SLAM_ICP_SPLIT / phase timing: last template argument 0 = ring search, 2 = list sweeps
        t = re.search(r"icp_fit_kernel<([^>]*)>", k)
        if t:
            name += "_list" if t.group(1).split(",")[-1].strip() == "2" else "_ring"
    return name


stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if stats:
    rows = list(csv.DictReader(open(stats[-1])))     # (gpurun merges into gpurun_out/: the newest run's file)
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, tag + "_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for k in sorted(acc):
        for c, v in sorted(acc[k].items()):
            w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])

traffic = {}
for k, cs in acc.items():
    if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
        fetch = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [0])), 1)
        write = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [0])), 1)
        traffic[k] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                      "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                      "note": "read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = 1/2 of a wide "
                              "coalesced read); Infinity-Cache hits are included in these counters"}
# how busy the vector ALUs were: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the wavefronts, one VALU
# instruction of a wavefront holds its SIMD for one quad-cycle (MI355X_MICROARCH.md, PMC units); GRBM_GUI_ACTIVE is
# summed over the 8 XCDs; 1024 SIMDs.  1.0 = every SIMD issued a VALU instruction in every cycle of the kernel.
for k, cs in acc.items():
    if "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
        mean = lambda c: sum(cs[c]) / len(cs[c])
        cycles = mean("GRBM_GUI_ACTIVE") / 8.0
        if cycles > 0:
            traffic.setdefault(k, {})["valu_busy_frac"] = 4.0 * mean("SQ_ACTIVE_INST_VALU") / (1024.0 * cycles)
            traffic[k]["kernel_cycles"] = cycles
    if "SQ_THREAD_CYCLES_VALU" in cs and "SQ_ACTIVE_INST_VALU" in cs:
        # lanes that did work per issued VALU instruction slot (64 = every lane of every instruction)
        mean = lambda c: sum(cs[c]) / len(cs[c])
        if mean("SQ_ACTIVE_INST_VALU") > 0:
            traffic.setdefault(k, {})["valu_active_lane_share"] = mean("SQ_THREAD_CYCLES_VALU") / (64.0 * mean("SQ_ACTIVE_INST_VALU"))
    if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs:
        mean = lambda c: sum(cs[c]) / len(cs[c])
        if mean("SQ_LDS_IDX_ACTIVE") > 0:
            traffic.setdefault(k, {})["lds_bank_conflict_share"] = mean("SQ_LDS_BANK_CONFLICT") / mean("SQ_LDS_IDX_ACTIVE")
json.dump(traffic, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1, sort_keys=True)
for name in ("bench.json", "config3.json", "config4.json", "config5.json", "build.txt", "two_ranks_one_gpu.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, tag + "_" + name))
# the side runs: kernel stats of the merged raycast, config 3 and the index build; counters of the merged raycast
for sub, out in (("stats_merge", "raycast_merge_kernel_stats.csv"), ("stats_c3", "config3_kernel_stats.csv"), ("stats_build", "build_kernel_stats.csv")):
    st = sorted(glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if st:
        rows = list(csv.DictReader(open(st[-1])))
        with open(os.path.join(dst, tag + "_" + out), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows:
                w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
macc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_merge", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        macc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if macc:
    with open(os.path.join(dst, tag + "_pmc_raycast_merge.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for k in sorted(macc):
            for c, v in sorted(macc[k].items()):
                w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
print("wrote", sorted(os.listdir(dst)))
