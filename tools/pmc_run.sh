#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes (separate rocprofv3 --pmc runs, kernel trace only) of one bench step for a library.
# usage: tools/pmc_run.sh tag lib.so [filter]
TAG=$1; LIB=$2; FLT=${3:-mbx}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export UDA_LIB=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_pmc$i -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side > $ROOT/gpurun_out/${TAG}_pmc$i.log 2>&1
  F=$(find $ROOT/gpurun_out/${TAG}_pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $ROOT/tools/pmc_table.py $F $FLT > $ROOT/gpurun_out/${TAG}_pmc$i.txt; rm -rf $ROOT/gpurun_out/${TAG}_pmc$i; else echo "no counters for set $i"; tail -3 $ROOT/gpurun_out/${TAG}_pmc$i.log; fi
done
cat $ROOT/gpurun_out/${TAG}_pmc*.txt
