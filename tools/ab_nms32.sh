#!/bin/bash
# Run ON THE GPU BOX: A/B of NMS kernel variants (alternative libraries) at batch 32: headline (cooperative kernel), spread scores
# (score prefix -> one-block kernel), D2 per-class.  usage: tools/ab_nms32.sh lib1.so lib2.so ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for LIB in "$@"; do
  export UDA_LIB=$ROOT/$LIB
  echo "== $(basename $LIB .so)"
  for ARGS in "" "--cls-spread 20" "--config 4"; do
    (cd $ROOT && timeout -k 10 300 python bench.py $ARGS --steps 5 --warmup 2 --no-side --no-cpu-baseline --protocol serial 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  [$ARGS] serial', d['ms_per_step'], 'nms', d['kernel_ms_per_step']['nms'])")
  done
done
