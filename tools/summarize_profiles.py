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

def csrc_sha():
    """sha256 over the kernel sources the profile was taken from (bench.py compares it with the sources it runs on)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def collect(pattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, pattern, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def write_pmc(acc, path):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for k in sorted(acc):
            for c, v in sorted(acc[k].items()):
                w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])


def traffic_of(acc):
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
        mean = lambda c: sum(cs[c]) / len(cs[c])
        if "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
            cycles = mean("GRBM_GUI_ACTIVE") / 8.0
            if cycles > 0:
                traffic.setdefault(k, {})["valu_busy_frac"] = 4.0 * mean("SQ_ACTIVE_INST_VALU") / (1024.0 * cycles)
                traffic[k]["kernel_cycles"] = cycles
        if "SQ_THREAD_CYCLES_VALU" in cs and "SQ_ACTIVE_INST_VALU" in cs:
            # lanes that did work per issued VALU instruction slot (64 = every lane of every instruction)
            if mean("SQ_ACTIVE_INST_VALU") > 0:
                traffic.setdefault(k, {})["valu_active_lane_share"] = mean("SQ_THREAD_CYCLES_VALU") / (64.0 * mean("SQ_ACTIVE_INST_VALU"))
        if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs:
            if mean("SQ_LDS_IDX_ACTIVE") > 0:
                traffic.setdefault(k, {})["lds_bank_conflict_share"] = mean("SQ_LDS_BANK_CONFLICT") / mean("SQ_LDS_IDX_ACTIVE")
        if "TCC_EA0_ATOMIC_sum" in cs:
            traffic.setdefault(k, {})["atomic_line_requests_per_launch"] = mean("TCC_EA0_ATOMIC_sum")
    traffic["_meta"] = {"tag": tag, "csrc_sha256": csrc_sha(),
                        "what": "csrc_sha256 = sha256 over slam_amd/csrc/*.hip, *.hpp as they were when this profile was summarised: "
                                "bench.py compares it with the sources it runs on and says `stale` in its roofline.valu block"}
    return traffic


def write_stats(sub, out):
    st = sorted(glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if not st:
        return
    rows = list(csv.DictReader(open(st[-1])))
    with open(os.path.join(dst, tag + "_" + out), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


acc = collect("pmc_[0-9]*")
write_pmc(acc, os.path.join(dst, tag + "_pmc.csv"))
json.dump(traffic_of(acc), open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1, sort_keys=True)
# the point-to-line solver's passes (bench.py --mode p2l) and the endpoint leg's
acc_p2l = collect("pmcp2l_[0-9]*")
if acc_p2l:
    write_pmc(acc_p2l, os.path.join(dst, tag + "_p2l_pmc.csv"))
    json.dump(traffic_of(acc_p2l), open(os.path.join(dst, tag + "_p2l_traffic.json"), "w"), indent=1, sort_keys=True)
acc_ep = collect("pmcep_[0-9]*")
if acc_ep:
    json.dump(traffic_of(acc_ep), open(os.path.join(dst, tag + "_endpoints_traffic.json"), "w"), indent=1, sort_keys=True)
write_stats("stats_p2l", "p2l_kernel_stats.csv")
write_stats("stats_endpoints", "endpoints_kernel_stats.csv")
for name in ("bench.json", "config3.json", "config4.json", "config5.json", "build.txt", "two_ranks_one_gpu.json", "force_dist.json",
             "force_dist_no_merge.json", "force_dist_cu_cap1.json", "p2l.json", "endpoints.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, tag + "_" + name))
        if name.endswith(".json"):
            # a line of this round was taken BEFORE the counter summary it stands beside existed: its `stale` compared the run's
            # sources with the summary committed then.  Restated against the summary written above (same tag): stale = the run's own
            # sources (run_csrc_sha256, recorded by bench.py) differ from the ones this summary is stamped with
            def restate(o):
                if isinstance(o, dict):
                    if "run_csrc_sha256" in o and "stale" in o:
                        o["profile_csrc_sha256"] = csrc_sha()
                        o["stale"] = o["run_csrc_sha256"] != csrc_sha()
                    # ... and the counter-derived figures of a roofline block are those of the summary written above, not of the
                    # one that was committed when the line was taken
                    v_ = o.get("valu")
                    if isinstance(v_, dict) and "kernel" in o and v_.get("source") and v_.get("run_csrc_sha256") == csrc_sha():
                        tf = os.path.join(dst, os.path.basename(v_["source"]))
                        k_ = json.load(open(tf)).get(o["kernel"].split(" ")[0], {}) if os.path.exists(tf) else {}
                        busy, lanes = k_.get("valu_busy_frac"), k_.get("valu_active_lane_share")
                        if busy is not None and lanes is not None:
                            held = v_.get("cus_held_by_one_launch") or 256
                            v_.update(busy_frac=busy, active_lane_share=lanes, frac=busy * lanes, frac_on_held_cus=busy * lanes * 256 / held)
                            o["traffic"] = k_.get("hbm_bytes_per_launch")
                    for v in o.values():
                        restate(v)
                elif isinstance(o, list):
                    for v in o:
                        restate(v)
            try:
                d = json.load(open(os.path.join(dst, tag + "_" + name)))
                restate(d)
                json.dump(d, open(os.path.join(dst, tag + "_" + name), "w"))
            except ValueError:
                pass
# the side runs: kernel stats of the merged raycast, config 3 and the index build; counters of the merged raycast
for sub, out in (("stats_merge", "raycast_merge_kernel_stats.csv"), ("stats_c3", "config3_kernel_stats.csv"), ("stats_build", "build_kernel_stats.csv"),
                 ("stats_c3cpp", "config3_cpp_kernel_stats.csv")):
    st = sorted(glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if st:
        rows = list(csv.DictReader(open(st[-1])))
        with open(os.path.join(dst, tag + "_" + out), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows:
                w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
# round 6: launches per match of the C++ adapter's default path, from the kernel trace of the binary (2 passes x 19 matches; the
# target is replaced every 10 clouds: its launches -- index build, bin order -- are told apart by their count)
tr = sorted(glob.glob(os.path.join(src, "stats_c3cpp", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if tr:
    rows = list(csv.DictReader(open(tr[-1])))
    fits = sum(int(r["Calls"]) for r in rows if "icp_fit_spread_kernel" in r["Name"])
    per_match, elsewhere = {}, {}
    for r in rows:
        (per_match if int(r["Calls"]) >= fits else elsewhere)[kname(r["Name"])] = (per_match if int(r["Calls"]) >= fits else elsewhere).get(kname(r["Name"]), 0) + int(r["Calls"])
    try:
        line = json.loads([l for l in open(os.path.join(src, "config3_cpp_profiled.json")) if l.startswith("{")][-1])
    except Exception:
        line = None
    json.dump({"matches_traced": fits,
               "launches_per_match": round(sum(per_match.values()) / max(fits, 1), 2),
               "kernels_launched_at_least_once_per_match": per_match,
               "kernels_of_the_target_updates_and_start_up": elsewhere,
               "runtime_copy_or_fill_kernels_per_match": round(sum(v for k, v in per_match.items() if "rocclr" in k) / max(fits, 1), 2),
               "line_of_the_traced_run": line,
               "what": "rocprofv3 --kernel-trace --stats of tests/cpp/ccicp_sequence (form seq: setSceneCloud + doICPMatch per cloud, "
                       "20 clouds, target replaced every 10, two passes); a kernel counts towards a match when it was launched at least "
                       "once per match"},
              open(os.path.join(dst, tag + "_config3_launches.json"), "w"), indent=1)
for name, out in (("spread_time.json", "spread_time.json"), ("spread_stamps.json", "spread_stamps.json"), ("latency_chase.txt", "latency_chase.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, tag + "_" + out))
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
# registers / scratch / LDS per kernel of the objects as they are built NOW (the last step: the table must describe what ships)
import subprocess
regs = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_regs.py")], capture_output=True, text=True)
if regs.returncode == 0 and regs.stdout:
    open(os.path.join(dst, tag + "_kernel_registers.txt"), "w").write(
        "# python tools/kernel_regs.py (llvm-readelf --notes of slam_amd/lib/obj/*.o), csrc sha256 %s\n%s" % (csrc_sha(), regs.stdout))


def design_table():
    """profiles/<tag>_design_table.md: the numbers of DESIGN.md section 4, one row per kernel of the default bench line, made from
    the files written above and nothing else (time from the kernel trace of the pipelined run, registers from the code objects,
    counters from the --pmc passes in which each kernel runs alone on the chip).  DESIGN.md includes it verbatim
    (tests/test_design_table.py checks that it does): a figure in that table cannot be older than its profile."""
    want = ["icp_fit_pair_kernel", "icp_fit_fused_kernel", "icp_fit_spread_kernel", "beams_from_scans_kernel", "tile_items_wg_kernel",
            "raycast_tiled_kernel", "endpoints_kernel", "finalize_reset_rows_kernel", "finalize_kernel"]
    alg = {  # SURVEY 8(d) algorithmic bytes per launch of the default bench line (config 2), where the kernel has a byte figure
        "icp_fit_pair_kernel": ("16*P + S*(8*M + 96)", 16 * 276242 + 256 * (8 * 10000 + 96)),
    }
    stats = {}
    for which, fn in (("", "_kernel_stats.csv"), ("p2l", "_p2l_kernel_stats.csv"), ("ep", "_endpoints_kernel_stats.csv")):
        f = os.path.join(dst, tag + fn)
        if os.path.exists(f):
            for r in csv.DictReader(open(f)):
                stats.setdefault(which, {}).setdefault(r["kernel"], r)
    regs = {}
    f = os.path.join(dst, tag + "_kernel_registers.txt")
    if os.path.exists(f):
        for line in open(f):
            m = re.match(r"(\w+)\s+(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s*$", line)
            if m:
                base = re.match(r"(\w+)", m.group(2)).group(1)
                v, vs, sg, ss, scr, lds = (int(m.group(i)) for i in range(3, 9))
                e = regs.setdefault(base, {"vgpr": [], "sgpr": [], "sspill": [], "scratch": []})
                e["vgpr"].append(v), e["sgpr"].append(sg), e["sspill"].append(ss), e["scratch"].append(scr)
    rng = lambda v: ("%d" % v[0]) if min(v) == max(v) else "%d-%d" % (min(v), max(v))
    rows = []

    def row(label, k, st, tr):
        r_ = st.get(k)
        t_ = tr.get(k, {})
        g_ = regs.get(k)
        busy, lanes = t_.get("valu_busy_frac"), t_.get("valu_active_lane_share")
        a_ = alg.get(k)
        avg_ms = float(r_["avg_ns"]) * 1e-6 if r_ else None
        rows.append("| `%s`%s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            k, label, r_["calls"] if r_ else "-", ("%.1f" % (float(r_["avg_ns"]) / 1e3)) if r_ else "-",
            ("%s / %s (%s spilled) / %s B" % (rng(g_["vgpr"]), rng(g_["sgpr"]), rng(g_["sspill"]), rng(g_["scratch"]))) if g_ else "-",
            ("%.1f %%" % (100 * busy)) if busy is not None else "-", ("%.1f %%" % (100 * lanes)) if lanes is not None else "-",
            ("%.1f %%" % (100 * busy * lanes)) if busy is not None and lanes is not None else "-",
            ("%.0f %%" % (100 * t_["lds_bank_conflict_share"])) if "lds_bank_conflict_share" in t_ else "-",
            ("%.2f MB" % (t_["hbm_bytes_per_launch"] / 1e6)) if "hbm_bytes_per_launch" in t_ else "-",
            ("%.1f MB = %s -> %.1f GB/s = %.4f of 8 TB/s" % (a_[1] / 1e6, a_[0], a_[1] / avg_ms / 1e6, a_[1] / avg_ms / 1e6 / 8000.0)) if a_ and avg_ms else "-"))

    tr = json.load(open(os.path.join(dst, tag + "_traffic.json"))) if os.path.exists(os.path.join(dst, tag + "_traffic.json")) else {}
    for k in want:
        if k in stats.get("", {}) or k in tr:
            row("", k, stats.get("", {}), tr)
    f = os.path.join(dst, tag + "_p2l_traffic.json")
    if os.path.exists(f) and "p2l" in stats:
        row(" (`--mode p2l`)", "icp_fit_pair_kernel", stats["p2l"], json.load(open(f)))
    f = os.path.join(dst, tag + "_endpoints_traffic.json")
    if os.path.exists(f) and "ep" in stats:
        row(" (endpoint leg alone)", "endpoints_kernel", stats["ep"], json.load(open(f)))
    head = ("<!-- generated by tools/summarize_profiles.py %s from profiles/%s_*; do not edit: regenerate -->\n"
            "| kernel | launches in the trace | avg us per launch (pipelined mix) | VGPR / SGPR / scratch (code objects) | VALU busy (alone) | active lanes | "
            "= of the chip's lane-slots | LDS bank-conflict share | HBM traffic per launch (counters) | algorithmic bytes (SURVEY 8d) -> of the HBM peak |\n"
            "|---|---|---|---|---|---|---|---|---|---|\n" % (tag, tag))
    bench = os.path.join(dst, tag + "_bench.json")
    foot = ""
    if os.path.exists(bench):
        try:
            b = json.loads([l for l in open(bench) if l.startswith("{")][-1])
            foot = ("\nThe un-profiled line of the same build (`profiles/%s_bench.json`): **%.4f ms per step = %.1f M registered scan-points/s**, %s; "
                    "`roofline.frac` %.4f (%.1f GB/s of algorithmic bytes per launch of `%s`, %.4f ms per launch, HIP events in the run); "
                    "kernel sources sha256 `%s`.\n" % (tag, b["ms_per_step"], b["value"] / 1e6, "%.1f G cell-updates/s" % (b.get("grid_cell_updates_per_s", 0) / 1e9),
                                                        b["roofline"]["frac"], b["roofline"]["achieved"], b["roofline"]["kernel"], b["roofline"]["avg_launch_ms"], csrc_sha()[:16]))
        except Exception as ex:
            foot = "\n(bench line unreadable: %r)\n" % (ex,)
    open(os.path.join(dst, tag + "_design_table.md"), "w").write(head + "\n".join(rows) + "\n" + foot)


design_table()
print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
