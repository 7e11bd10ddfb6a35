"""Kernel-trace gaps of one batch-1 step: python tools/gap_probe.py kernel_trace.csv  (idle time between consecutive kernels)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]] if len(idx) > 1 else rows[idx[-1]:]
t0, busy, gaps, end = int(seg[0]["Start_Timestamp"]), 0, [], None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None:
        gaps.append((s - end, r["Kernel_Name"].split("(")[0][-30:]))
    busy += e - s
    end = max(end or 0, e)
span = end - t0
print("kernels %d, span %.1f us, busy %.1f us, idle %.1f us" % (len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3))
gaps.sort(reverse=True)
print("largest gaps (us):", [(round(g / 1e3, 1), n) for g, n in gaps[:12]])
print("median gap %.2f us" % (sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3))
