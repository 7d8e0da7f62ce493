import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])):
    print(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40], r["Calls"], r["AverageNs"])
