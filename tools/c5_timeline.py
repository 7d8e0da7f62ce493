"""Device timeline of a profiled run (rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --config 5 ...):
the registration launches with the gaps between them, and what else ran around one of them.
    python tools/c5_timeline.py gpurun_out/c5prof"""
import csv
import glob
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c5prof"
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return (m.group(1) if m else n.split("(")[0])[:28]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
icp = [e for e in ev if "icp_fit" in e[2]]
print(len(ev), "dispatches,", len(icp), "registration launches")
k0 = min(8, len(icp) - 2)
t0 = icp[k0][0]
print("registration launches from number %d: start, duration, gap to the next (us)" % k0)
for a, b in zip(icp[k0:k0 + 14], icp[k0 + 1:k0 + 15]):
    print("  start %8.1f dur %6.1f gap %6.1f  queue %s" % ((a[0] - t0) / 1e3, (a[1] - a[0]) / 1e3, (b[0] - a[1]) / 1e3, a[3]))
a, b = icp[k0 + 4], icp[k0 + 6]
print("everything else between %.1f and %.1f us:" % ((a[0] - t0) / 1e3, (b[1] - t0) / 1e3))
for e in ev:
    if e[1] > a[0] and e[0] < b[1] and "icp_fit" not in e[2]:
        print("   %-28s start %8.1f end %8.1f dur %6.1f queue %s" % (e[2], (e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3]))
print("compact view (I = registration, R = raycast, c = copyBuffer, f = fill, w = window_points, . = a rebuild kernel) from %.1f us:" % 0.0)
line = []
for e in ev:
    if e[0] < t0 or e[0] > t0 + 6000e3:
        continue
    n = e[2]
    tag = "I" if "icp_fit" in n else ("R" if "raycast" in n else ("c" if "copyBuffer" in n else ("f" if "fillBuffer" in n else ("w" if "window_points" in n else ("b" if "beams" in n or "tile_items" in n else ".")))))
    line.append("%s%.0f-%.0f" % (tag, (e[0] - t0) / 1e3, (e[1] - t0) / 1e3))
print(" ".join(line))
