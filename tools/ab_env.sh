#!/bin/bash
# Run ON THE GPU BOX: quick A/B of environment switches in one job (same box, same library):
#   tools/ab_env.sh "UDA_FUSE_IN=0" "UDA_FUSE_IN=1" "UDA_FUSE_IN=1 UDA_SEPF_ALL=1"
# two rounds, alternating; prints ms/step and the per-kind kernel times of each run ("-" = no switch).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for ROUND in 1 2; do
  for CFG in "$@"; do
    SETS=""; [ "$CFG" != "-" ] && SETS="$CFG"
    env $SETS timeout -k 10 200 python $ROOT/bench.py --steps 10 --warmup 3 --no-side --no-cpu-baseline 2> /tmp/ab_err.txt > /tmp/ab_out.json || { echo "$CFG FAILED"; tail -3 /tmp/ab_err.txt; exit 1; }
    python - <<PY
import json
d=json.load(open("/tmp/ab_out.json")); k=d["kernel_ms_per_step"]
print("%-36s %6.2f ms/step  mbx %.2f pw %.2f sep %.2f fuse %.2f nms %.2f stem %.2f" % ("$CFG", d["ms_per_step"], k["mbx"], k["pw"], k["sep"], k["fuse"], k["nms"], k["stem"]))
PY
  done
done
