#!/usr/bin/env python
"""gpurun_out/<tag>_* (tools/collect_r04.sh) -> profiles/<tag>_*  (round 4: the default scheme is two fp16 pieces, float32-class):
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --warmup 1`
  <tag>_per_op.txt         every op of the plan with its duration, algorithmic rate and PMC traffic (tools/analyze_trace.py)
  <tag>_traffic.json       HBM bytes per launch and kernel kind (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes)
  <tag>_sq.txt             SQ counters per kernel (four passes) + derived busy fractions - the evidence DESIGN.md argues from
  <tag>_bf16x3_kernel_stats.csv   kernel stats of the same command under UDA_PW_SCHEME=bf16x3 (three bf16 pieces, six cross terms)
  <tag>_bench.json, <tag>_summary.md"""
import csv
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda n: os.path.join(root, "gpurun_out", "%s_%s" % (tag, n))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
o = lambda n: os.path.join(out, "%s_%s" % (tag, n))

open(o("kernel_stats.csv"), "w").write(open(g("kernel_stats.csv")).read())
if os.path.exists(g("bf16x3_kernel_stats.csv")):
    open(o("bf16x3_kernel_stats.csv"), "w").write(open(g("bf16x3_kernel_stats.csv")).read())
subprocess.check_call([sys.executable, os.path.join(root, "tools", "traffic_from_pmc.py"), g("FETCH_SIZE.csv"), g("WRITE_SIZE.csv"),
                       o("traffic.json")])
t = json.load(open(o("traffic.json")))
json.dump({k: v["hbm_bytes_per_launch"] for k, v in t.items()}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
per_op = subprocess.run([sys.executable, os.path.join(root, "tools", "analyze_trace.py"), g("kernel_trace.csv"), "--chunk", "32",
                         "--fetch", g("FETCH_SIZE.csv"), "--write", g("WRITE_SIZE.csv"), "--top", "60"], capture_output=True, text=True)
open(o("per_op.txt"), "w").write(per_op.stdout + per_op.stderr[-2000:])
line = [l for l in open(g("bench.json")) if l.startswith("{")][-1]
open(o("bench.json"), "w").write(line)
b = json.loads(line)

# ---- SQ tables: merge the three passes per kernel, add the averages of the kernel stats and derived fractions
short = lambda n: n.replace("void ", "").replace("uda::", "").split("(")[0]
stats = {short(r["Name"]): r for r in csv.DictReader(open(g("kernel_stats.csv")))}
cnt = {}
PASSES = [i for i in (1, 2, 3, 4) if os.path.exists(g("sq%d.txt" % i)) and os.path.getsize(g("sq%d.txt" % i)) > 0]
for i in PASSES:
    rows = [l.rstrip("\n") for l in open(g("sq%d.txt" % i))]
    head = rows[0].split()
    cols = head[2:]
    for l in rows[1:]:
        name, rest = l[:34].strip().replace("uda::", ""), l[34:].split()
        if not rest:
            continue
        d = cnt.setdefault(name, {"n": int(rest[0])})
        for c, v in zip(cols, rest[1:]):
            d[c] = float(v)
full = {"ACTIVE_INST_VALU": "SQ_ACTIVE_INST_VALU", "_ACTIVE_INST_ANY": "SQ_ACTIVE_INST_ANY", "_ACTIVE_INST_LDS": "SQ_ACTIVE_INST_LDS",
        "DS_BANK_CONFLICT": "SQ_LDS_BANK_CONFLICT", "Q_LDS_IDX_ACTIVE": "SQ_LDS_IDX_ACTIVE", "MFMA_BUSY_CYCLES": "SQ_VALU_MFMA_BUSY_CYCLES",
        "READ_CYCLES_VALU": "SQ_THREAD_CYCLES_VALU", "_INSTS_VALU_TRANS": "SQ_INSTS_VALU_TRANS", "INSTS_VALU_TRANS": "SQ_INSTS_VALU_TRANS",
        "INST_CYCLES_SALU": "SQ_INST_CYCLES_SALU", "SQ_INSTS_VMEM_RD": "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR": "SQ_INSTS_VMEM_WR"}
with open(o("sq.txt"), "w") as f:
    f.write("# SQ counters per kernel, mean per launch (rocprofv3 --pmc, separate passes of `python bench.py --steps 1 --warmup 1\n"
            "# --no-side --no-cpu-baseline`, tools/collect_r04.sh; default scheme = two fp16 pieces), and what DESIGN.md derives from them.\n"
            "#   clock      = SQ_BUSY_CYCLES / 32 shader engines / kernel time (the clock the chip holds under that kernel)\n"
            "#   VALU busy  = 4 x SQ_ACTIVE_INST_VALU / (32 x SQ_BUSY_CYCLES)    (quad-cycles -> cycles; 1024 SIMDs = 32 SEs x 32)\n"
            "#   MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)\n"
            "#   LDS busy   = SQ_LDS_IDX_ACTIVE / (8 x SQ_BUSY_CYCLES)           (LDS-array cycles over 256 CUs)\n"
            "#   wave time  = issuing (SQ_ACTIVE_INST_ANY) / issue-stalled (SQ_WAIT_INST_ANY) / parked (SQ_WAIT_ANY), shares of SQ_WAVE_CYCLES\n"
            "#   per wave   = SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_INSTS_LDS divided by SQ_WAVES (SQ_INSTS_VALU_TRANS is not collected on this stack)\n\n")
    hdr = ("%-34s %3s %9s %6s | %5s %5s %5s | %5s %5s %5s | %7s %5s %5s | %9s\n" %
           ("kernel", "n", "avg us", "GHz", "VALU%", "MFMA%", "LDS%", "iss%", "stl%", "park%", "VALU/wv", "MFMA", "LDS", "bank cfl"))
    f.write(hdr + "-" * len(hdr) + "\n")
    order = sorted(cnt, key=lambda k: -float(stats.get(k, {}).get("TotalDurationNs", 0)) if k in stats else 0)
    for k in order:
        d = cnt[k]
        if "SQ_BUSY_CYCLES" not in d or k not in stats:
            continue
        us = float(stats[k]["AverageNs"]) / 1e3
        busy = d["SQ_BUSY_CYCLES"]
        ghz = busy / 32 / (us * 1e-6) / 1e9
        gv = lambda name: d.get(name, d.get(name[-16:], 0.0))
        wc = d.get("SQ_WAVE_CYCLES", 1.0)
        waves = d.get("SQ_WAVES", 1.0)
        f.write("%-34s %3d %9.1f %6.2f | %5.1f %5.1f %5.1f | %5.1f %5.1f %5.1f | %7.0f %5.0f %5.0f | %9.3g\n" % (
            k[:34], d["n"], us, ghz, 400 * gv("SQ_ACTIVE_INST_VALU") / (32 * busy), 100 * gv("SQ_VALU_MFMA_BUSY_CYCLES") / (32 * busy),
            100 * gv("SQ_LDS_IDX_ACTIVE") / (8 * busy), 100 * gv("SQ_ACTIVE_INST_ANY") / wc, 100 * gv("SQ_WAIT_INST_ANY") / wc,
            100 * gv("SQ_WAIT_ANY") / wc, gv("SQ_INSTS_VALU") / waves, gv("SQ_INSTS_MFMA") / waves,
            gv("SQ_INSTS_LDS") / waves, gv("SQ_LDS_BANK_CONFLICT")))
    f.write("\n# raw tables (tools/pmc_table.py)\n")
    for i in PASSES:
        f.write("\n## pass %d\n" % i + open(g("sq%d.txt" % i)).read())

rows = list(csv.DictReader(open(g("kernel_stats.csv"))))
with open(o("summary.md"), "w") as f:
    f.write("# %s profile summary (MI355X, `python bench.py`, BASELINE configs[1])\n\n" % tag)
    pr = b.get("precision", {})
    f.write("bench line: **%.1f %s**, %.2f ms/step, p50 %.2f ms (%s); fed inside every step %s ms pipelined / %s ms serial; batch-1 "
            "detect latency %s ms; other schemes: bf16x3 %s ms, bf16x2 %s ms, exact f32 MFMA %s ms; configs[2] / [3] / [4] %s / %s / %s ms; "
            "head-only MC %s ms; spread scores %s ms; process-group path (world 1) %s ms; roofline %s\n\n" % (
                b["value"], b["unit"], b["ms_per_step"], b.get("p50_step_ms", 0.0), b.get("dtype"),
                b.get("h2d_inclusive_pipelined", {}).get("ms_per_step"), b.get("h2d_inclusive_serial", {}).get("ms_per_step"),
                b.get("p50_detect_latency_ms"), pr.get("bf16x3", {}).get("ms_per_step"), pr.get("bf16x2", {}).get("ms_per_step"),
                pr.get("f32", {}).get("ms_per_step"), *[b.get("configs", {}).get(k, {}).get("ms_per_step") for k in ("2", "3", "4")],
                b.get("head_only", {}).get("ms_per_step"), b.get("nms_spread_scores", {}).get("ms_per_step"),
                b.get("rccl_world1", {}).get("ms_per_step"), json.dumps(b["roofline"])))
    f.write("cpu_baseline: %s\n\n" % json.dumps(b.get("cpu_baseline")))
    f.write("## rocprofv3 --kernel-trace --stats (3 timed + 2 warm-up steps; `%s_kernel_stats.csv`)\n\n" % tag)
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:26]:
        f.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (short(r["Name"])[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"]))
    fam = [r for r in rows if "mbx" in r["Name"]]
    calls = sum(int(r["Calls"]) for r in fam)
    tot = sum(float(r["TotalDurationNs"]) for r in fam) / 1e6
    f.write("\nfused MBConv family: %d launches, %.2f ms = %.2f ms per step (15 launches per step), %.3f ms per launch\n" % (
        calls, tot, tot / (calls / 15.0), tot / calls))
    f.write("\n## HBM traffic from PMC (`%s_traffic.json`; FETCH_SIZE x2 on gfx950, separate passes)\n\n" % tag)
    f.write("| kernel | launches | read MB/launch | write MB/launch |\n|---|---|---|---|\n")
    for k, v in t.items():
        f.write("| `%s` | %d | %.1f | %.1f |\n" % (k, v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
    f.write("\nSQ counter tables with derived busy fractions: `%s_sq.txt`; every op with its rate and traffic: `%s_per_op.txt`.\n" % (tag, tag))
print(open(o("summary.md")).read()[:2500])
print(open(o("sq.txt")).read()[:6000])
