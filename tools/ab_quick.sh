#!/bin/bash
# Run ON THE GPU BOX: quick A/B of alternative libraries in one job (same box): tools/ab_quick.sh lib1.so lib2.so ...
# two rounds, alternating; prints ms/step and the per-kind kernel times of each run.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for ROUND in 1 2; do
  for LIB in "$@"; do
    UDA_LIB=$ROOT/$LIB timeout -k 10 200 python $ROOT/bench.py --steps 10 --warmup 3 --no-side --no-cpu-baseline 2> /tmp/ab_err.txt > /tmp/ab_out.json || { echo "$LIB FAILED"; tail -3 /tmp/ab_err.txt; exit 1; }
    python - <<PY
import json
d=json.load(open("/tmp/ab_out.json")); k=d["kernel_ms_per_step"]
print("%-28s %6.2f ms/step  mbx %.2f pw %.2f sep %.2f fuse %.2f nms %.2f stem %.2f pre %.2f" % ("$LIB", d["ms_per_step"], k["mbx"], k["pw"], k["sep"], k["fuse"], k["nms"], k["stem"], k["preprocess"]))
PY
    grep "stamps" /tmp/ab_err.txt | tail -1
  done
done
