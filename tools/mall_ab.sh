#!/bin/bash
# Run ON THE GPU BOX: Infinity-Cache windows A/B in one job: tools/mall_ab.sh 0 120 160 200 240   (budgets in MB; 0 = off)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for ROUND in 1 2; do
  for MB in "$@"; do
    UDA_MALL_MB=$MB timeout -k 10 200 python $ROOT/bench.py --steps 10 --warmup 3 --no-side --no-cpu-baseline --protocol serial 2> /tmp/ab_err.txt > /tmp/ab_out.json || { echo "$MB FAILED"; tail -3 /tmp/ab_err.txt; exit 1; }
    python - <<PY
import json
d=json.load(open("/tmp/ab_out.json")); k=d["kernel_ms_per_step"]
print("UDA_MALL_MB=%-5s %6.2f ms/step  mbx %.2f pw %.2f se %.2f sep %.2f nms %.2f" % ("$MB", d["ms_per_step"], k["mbx"], k["pw"], k["se"], k["sep"], k["nms"]))
PY
  done
done
