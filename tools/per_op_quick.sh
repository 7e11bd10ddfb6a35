#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short bench run mapped onto the op list (tools/analyze_trace.py); env switches pass through.
#   tools/per_op_quick.sh [top N] [pmc]      pmc: FETCH_SIZE / WRITE_SIZE passes too (HBM-side traffic per op)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side --protocol serial"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- $B > /tmp/pq.log 2>&1 || { tail -5 /tmp/pq.log; exit 1; }
F=$(find /tmp/pq -name "*kernel_trace.csv" | head -1)
EXTRA=""
if [ "$2" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pq_$C -- $B > /tmp/pq_$C.log 2>&1 || { tail -5 /tmp/pq_$C.log; exit 1; }
  done
  EXTRA="--fetch $(find /tmp/pq_FETCH_SIZE -name '*counter_collection.csv' | head -1) --write $(find /tmp/pq_WRITE_SIZE -name '*counter_collection.csv' | head -1)"
fi
cd $ROOT && python tools/analyze_trace.py $F --chunk 32 --top ${1:-70} $EXTRA > $OUT/per_op_quick.txt
rm -rf /tmp/pq /tmp/pq_FETCH_SIZE /tmp/pq_WRITE_SIZE
grep -E "sep|fuse" $OUT/per_op_quick.txt
