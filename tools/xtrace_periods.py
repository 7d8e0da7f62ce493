"""Per-chunk host periods of a slam_mapper_t from its call trace (measurement build only: python -m slam_amd.build --measure,
then SLAM_AMD_MEASURE=1 SLAM_MAPPER_TRACE=gpurun_out/xtrace.txt python bench.py --config 5): where the producer's time goes
between two pushes, rebuild pushes apart from the others.    python tools/xtrace_periods.py [gpurun_out/xtrace.txt]"""
import sys
L = [l.split() for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/xtrace.txt")]
ev = [(" ".join(x[:-2]), float(x[-2])) for x in L]
pushes = [i for i, (n, t) in enumerate(ev) if n == "push"]
per_reb, per_no = [], []
for k, (a, b) in enumerate(zip(pushes, pushes[1:])):
    seg = ev[a:b]
    names = [n for n, _ in seg]
    d = {}
    for (n, t), (n2, t2) in zip(seg, seg[1:] + [ev[b]]):
        d[n2] = d.get(n2, 0) + (t2 - t)
    per = ev[b][1] - ev[a][1]
    if k >= 12:
        (per_reb if "begun" in names else per_no).append(per)
    print("%2d period %6.1f %s  %s" % (k, per, "REBUILD" if "begun" in names else "       ", " ".join("%s=%.0f" % (k2, v) for k2, v in d.items() if v > 30)))
import statistics as S
print("from chunk 12: rebuild pushes %.1f us (n=%d), others %.1f us (n=%d)" % (S.mean(per_reb), len(per_reb), S.mean(per_no), len(per_no)))
