#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: A/B of alternative builds of libuda_hip.so in ONE job
# (timings from different boxes differ by several per cent).  usage: tools/ab_run.sh tag lib1.so lib2.so ...
# For every library: smoke parity check, bench line (no CPU baseline) and a rocprofv3 kernel-stats summary.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  export UDA_LIB=$ROOT/$LIB
  echo "== $NAME"
  (cd $ROOT && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()") > $ROOT/gpurun_out/${TAG}_${NAME}_smoke.log 2>&1 || { echo "smoke FAILED for $NAME"; tail -5 $ROOT/gpurun_out/${TAG}_${NAME}_smoke.log; continue; }
  (cd $ROOT && timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side) > $ROOT/gpurun_out/${TAG}_${NAME}_bench.json 2> $ROOT/gpurun_out/${TAG}_${NAME}_bench.err
  python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/${TAG}_${NAME}_bench.json"))
print("$NAME", d["ms_per_step"], d["kernel_ms_per_step"])
PY
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_${NAME}_stats -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side) > $ROOT/gpurun_out/${TAG}_${NAME}_stats.log 2>&1
  F=$(find $ROOT/gpurun_out/${TAG}_${NAME}_stats -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $ROOT/gpurun_out/${TAG}_${NAME}_kernel_stats.csv && rm -rf $ROOT/gpurun_out/${TAG}_${NAME}_stats
done
echo done
