#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run: usage: kstats_top.py <dir> [n=25]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print("%6.2f%% %8.2f ms %6d x %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6,
                                                 int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("total %.1f ms" % (tot / 1e6))
