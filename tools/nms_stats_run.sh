#!/bin/bash
# Run ON THE GPU BOX with the -DUDA_NMS_STATS build (tools/libuda_nmsstats.so): phase times, list sizes and winners per
# grid-wide step of nms_coop_kernel for the headline batch and for batch 1, with one winner per step and with the default.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for W in 1 0 8; do
  for B in 32 1; do
    echo "=== UDA_NMS_WINNERS=$W batch $B"
    UDA_NMS_DEBUG=1 UDA_NMS_WINNERS=$W UDA_LIB=$ROOT/tools/libuda_nmsstats.so timeout -k 10 200 python $ROOT/bench.py --batch $B --steps 3 --warmup 1 --no-side --no-cpu-baseline --protocol serial 2> /tmp/st_err.txt > /tmp/st_out.json || { echo FAILED; tail -5 /tmp/st_err.txt; }
    grep -E "winners|nms phases|nms lists|cooperative NMS: [0-9]" /tmp/st_err.txt | tail -4
    python -c "import json; d=json.load(open('/tmp/st_out.json')); print('ms/step', d['ms_per_step'], 'nms', d['kernel_ms_per_step']['nms'])"
  done
done
