import csv, glob, re, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
icp = [e for e in ev if "icp_fit_fused" in e[2] or "icp_fit_pair" in e[2]]
icp = icp[6:]   # past the warm-up
dur = [(e[1]-e[0])/1e3 for e in icp]
gap = [(b[0]-a[1])/1e3 for a, b in zip(icp, icp[1:])]
per = [(b[0]-a[0])/1e3 for a, b in zip(icp, icp[1:])]
import statistics as S
print(d, "fits", len(icp), "dur avg %.1f med %.1f" % (S.mean(dur), S.median(dur)), "gap avg %.1f med %.1f max %.1f" % (S.mean(gap), S.median(gap), max(gap)), "period avg %.1f" % S.mean(per))
big = sorted(gap)[-12:]
print("  largest gaps:", [round(g) for g in big])
# first vs last quarter durations
q = len(dur)//4
print("  dur first quarter %.1f last quarter %.1f" % (S.mean(dur[:q]), S.mean(dur[-q:])))
