#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: everything profiles/r04_* is made from, for the default bench command
# (main leg = the shipped scheme, UDA_PW_SCHEME=f16x2: two fp16 pieces per operand, float32-class products).
#   kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md), four SQ counter passes over
#   ALL kernels (tools/pmc_table.py tables), the bench line itself, and the kernel stats of the bf16x3 (six-term) scheme beside it.
#   tools/summarize_r04.py (run afterwards, anywhere) turns gpurun_out/<tag>_* into profiles/<tag>_*.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --protocol serial"      # (one step at a time: no post-process of the previous step beside the first ops)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $B > $OUT/${TAG}_stats.log 2>&1
for f in kernel_stats kernel_trace; do F=$(find $OUT/${TAG}_stats -name "*${f}.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${TAG}_${f}.csv; done
rm -rf $OUT/${TAG}_stats
echo "stats done"
UDA_PW_SCHEME=bf16x3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats6 -- $B > $OUT/${TAG}_stats6.log 2>&1
F=$(find $OUT/${TAG}_stats6 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${TAG}_bf16x3_kernel_stats.csv
rm -rf $OUT/${TAG}_stats6
echo "bf16x3 stats done"
B1="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side --protocol serial"
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
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${TAG}_sq$i -- $B1 > $OUT/${TAG}_sq$i.log 2>&1
  F=$(find $OUT/${TAG}_sq$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $ROOT/tools/pmc_table.py $F "" > $OUT/${TAG}_sq$i.txt; else echo "no counters for set $i"; tail -3 $OUT/${TAG}_sq$i.log; fi
  rm -rf $OUT/${TAG}_sq$i
  echo "sq$i done"
done
cd $ROOT && python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo collected $TAG
