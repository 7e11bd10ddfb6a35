#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 --pmc counters: python tools/pmc_table.py counter_collection.csv [filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void uda::", "")
    if flt and flt not in name:
        continue
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        cnt[name] += 1
counters = sorted({c for v in agg.values() for c in v})
print("%-34s %5s " % ("kernel", "n") + " ".join("%16s" % c[-16:] for c in counters))
for name in sorted(agg, key=lambda n: -agg[n].get(counters[0], 0)):
    print("%-34s %5d " % (name[-34:], cnt[name]) + " ".join("%16.4g" % (agg[name].get(c, 0) / cnt[name]) for c in counters))
