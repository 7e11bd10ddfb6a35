#!/bin/bash
# A/B of libraries at batch 1 and at the default batch: tools/ab_b1.sh lib1.so lib2.so ...
for LIB in "$@"; do export UDA_LIB=$GRAFT_REPO_ROOT/$LIB
  python bench.py --batch 1 --chunk 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$LIB b1', d['ms_per_step'], d['kernel_ms_per_step'])"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$LIB b32', d['ms_per_step'], d['kernel_ms_per_step']['se'])"
done
