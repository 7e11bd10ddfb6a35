#!/usr/bin/env python
"""Compare rocprofv3 kernel_stats.csv files of A/B builds: per kernel name, average microseconds per launch.
python tools/ab_compare.py a_kernel_stats.csv b_kernel_stats.csv ..."""
import csv
import sys

tabs = []
for f in sys.argv[1:]:
    tabs.append({r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))})
names = sorted(set().union(*tabs), key=lambda n: -max(t.get(n, (0, 0))[1] for t in tabs))
print("%-70s" % "kernel" + "".join("%14s" % f.split("/")[-1][:13] for f in sys.argv[1:]))
for n in names[:40]:
    row = "%-70s" % n.replace("uda::", "").replace("void ", "")[:69]
    for t in tabs:
        c, tot = t.get(n, (0, 0))
        row += "%8.1f x%-4d" % (tot / max(c, 1) / 1e3, c)
    print(row)
print("%-70s" % "TOTAL ms" + "".join("%14.2f" % (sum(v[1] for v in t.values()) / 1e6) for t in tabs))
