#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: the rocprofv3 passes profiles/<tag>_* are made from, for ANY bench
# configuration (round 5: the headline, `--variant head` = what the reference's shipped inference YAMLs run, `--config 4` = D2).
#   tools/collect_r05.sh <tag> "<extra bench.py args>" [full|lite]
#   full: kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes, four SQ passes, the bench line;  lite: stats, FETCH / WRITE, SQ pass 1.
# Counters are collected in their own runs (--pmc with --kernel-trace only), FETCH_SIZE is doubled on gfx950 by the summariser
# (MI355X_MICROARCH.md).  tools/summarize_r05.py <tag> turns gpurun_out/<tag>_* into profiles/<tag>_*.
TAG=${1:-r05}
ARGS=${2:-}
MODE=${3:-full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-side --protocol serial"      # (one step at a time: no post-process of the previous step beside the first ops)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $B > $OUT/${TAG}_stats.log 2>&1
for f in kernel_stats kernel_trace; do F=$(find $OUT/${TAG}_stats -name "*${f}.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${TAG}_${f}.csv; done
rm -rf $OUT/${TAG}_stats
echo "stats done"
B1="python $ROOT/bench.py $ARGS --steps 1 --warmup 1 --no-cpu-baseline --no-side --protocol serial"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_$C -- $B1 > $OUT/${TAG}_$C.log 2>&1
  F=$(find $OUT/${TAG}_$C -name "*counter_collection.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${TAG}_${C}.csv
  rm -rf $OUT/${TAG}_$C
  echo "$C done"
done
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES_EQ_64"; do
  i=$((i+1))
  if [ "$MODE" = "lite" ] && [ $i -gt 1 ]; then break; fi
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${TAG}_sq$i -- $B1 > $OUT/${TAG}_sq$i.log 2>&1
  F=$(find $OUT/${TAG}_sq$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $ROOT/tools/pmc_table.py $F "" > $OUT/${TAG}_sq$i.txt; else echo "no counters for set $i"; tail -3 $OUT/${TAG}_sq$i.log; fi
  rm -rf $OUT/${TAG}_sq$i
  echo "sq$i done"
done
cd $ROOT && python bench.py $ARGS --steps 20 --warmup 5 $( [ "$MODE" = "lite" ] && echo "--no-side --no-cpu-baseline" ) > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo collected $TAG
