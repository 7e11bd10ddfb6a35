#!/bin/bash
# Run ON THE GPU BOX: the bounded Infinity-Cache experiment of VERDICT r02 (Next 9).  Blocks 6-15 (or a sub-range) run
# block by block on a few images at a time (fused front half -> SE -> projection), everything else on the full batch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { python bench.py --steps 6 --warmup 2 --child --no-side --no-cpu-baseline 2>/dev/null | tail -1; }
echo "baseline"; run
for cfg in "6-15 2" "6-15 4" "12-15 4" "12-15 8" "9-15 2" "6-10 2" "6-10 4" "12-15 16"; do
  set -- $cfg
  echo "UDA_MALL_BLOCKS=$1 UDA_MALL_IMAGES=$2"
  UDA_MALL_BLOCKS=$1 UDA_MALL_IMAGES=$2 run
done
