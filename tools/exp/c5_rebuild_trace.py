"""Round 5: what a rebuild of config 5's sliding target looks like on the device, from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --config 5 ...):  python tools/exp/c5_rebuild_trace.py DIR [n]
prints, for the n-th rebuild (default: the one in the middle), every kernel from its first thinning kernel to its last list
kernel -- start offset, duration, queue, grid -- and the registrations that ran meanwhile."""
import csv, glob, os, re, sys


def short(raw):
    n = raw.replace("(anonymous namespace)::", "").replace("void ", "")
    tag = "/W" if "WindowSource" in raw else ("/C" if "ChunkSource" in raw else "")
    return re.split(r"[<(]", n)[0].split("::")[-1] + tag


d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else None
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
      int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)) for r in rows]
K.sort()
starts = [i for i, k in enumerate(K) if k[2] in ("set_counts_kernel", "thin_min_kernel/W")]
print("%d kernels, %d rebuilds" % (len(K), len(starts)))
if not starts:
    sys.exit(0)
i0 = starts[which if which is not None else len(starts) // 2]
t0 = K[i0][0]
# the rebuild ends with its last list/normal kernel: take everything within 3 ms and cut at the last build kernel
build_names = ("thin_", "set_counts", "idx_", "scan_", "plan_", "list_", "p2l_", "extent", "bbox", "fill", "copyBuffer", "Memset", "fillBuffer")
win = [k for k in K[i0:] if k[0] - t0 < 3_000_000]
last = max(j for j, k in enumerate(win) if any(b in k[2] for b in build_names) and "icp_fit" not in k[2])
prev_end = None
for k in win[:last + 1]:
    is_build = any(b in k[2] for b in build_names)
    print("%9.1f us  %7.1f us  q%-3s %-34s grid %7d / %4d %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2][:34], k[4], k[5],
          ("  (+%.1f since the build's last kernel ended)" % ((k[0] - prev_end) / 1e3)) if (is_build and prev_end) else ""))
    if is_build:
        prev_end = k[1]
print("rebuild: %.1f us from first to last kernel" % ((win[last][1] - t0) / 1e3))
