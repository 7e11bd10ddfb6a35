#!/bin/bash
# Run ON THE GPU BOX: A/B of NMS kernel variants (alternative libraries) at small batches, one job.  usage: tools/ab_nms.sh lib1.so lib2.so ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  export UDA_LIB=$ROOT/$LIB
  echo "== $NAME"
  (cd $ROOT && timeout -k 10 200 python tools/debug/latency_probe.py 2>/dev/null | grep -o "^[a-z]* p50 wall [0-9.]* ms\|'nms': ([0-9.]*" | tr '\n' ' '; echo)
  for B in 4 8; do
    (cd $ROOT && timeout -k 10 200 python bench.py --batch $B --steps 5 --warmup 2 --no-side --no-cpu-baseline --protocol serial 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $B', d['ms_per_step'], 'nms', d['kernel_ms_per_step']['nms'])")
  done
done
