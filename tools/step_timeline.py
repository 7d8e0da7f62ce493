"""Device timeline of the pipelined bench from a kernel trace (rocprofv3 --kernel-trace --output-format csv -d DIR -- python3
bench.py --steps K ...): the registration launches of the timed region (the last K pair launches before the one-stream
comparison), their gaps per stream, and the span from the first to the end of the last grid kernel.
    python tools/step_timeline.py DIR K W      (K timed steps after W warm-up steps)"""
import csv
import glob
import re
import sys

d, K, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return (m.group(1) if m else n.split("(")[0])[:28]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
pair = [e for e in ev if e[2] == "icp_fit_pair_kernel"]
timed = pair[W:W + K]                  # warm-up launches come first; the event-bracketed runs behind the timed region use it too
t0 = timed[0][0]
print("%d pair launches in the trace; the timed region's %d:" % (len(pair), K))
for e in timed:
    print("  start %8.1f end %8.1f dur %6.1f queue %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3]))
last_icp = max(e[1] for e in timed)
grid = [e for e in ev if e[0] >= t0 - 1000 and e[2] in ("raycast_tiled_kernel", "finalize_reset_rows_kernel", "beams_from_scans_kernel", "tile_items_wg_kernel")]
fins = [e for e in grid if e[2] == "finalize_reset_rows_kernel"][:K]
print("last registration ends at %.1f us; the K-th finalize_reset ends at %.1f us -> %.4f ms per step on the device" %
      ((last_icp - t0) / 1e3, (fins[-1][1] - t0) / 1e3, (fins[-1][1] - t0) / 1e6 / K))
rc = [e for e in grid if e[2] == "raycast_tiled_kernel"][:K]
print("raycast durations (us):", " ".join("%.0f" % ((e[1] - e[0]) / 1e3) for e in rc))
