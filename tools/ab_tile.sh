for t in 412 78; do export UDA_MBXB_S2_TILE=$t UDA_LIB=$GRAFT_REPO_ROOT/ab/libuda_t$t.so
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t${t}_smoke.log 2>&1 || { echo smoke failed $t; tail -5 gpurun_out/t${t}_smoke.log; }
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile', '$t', d['ms_per_step'], d['kernel_ms_per_step'])"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t${t}_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > /dev/null 2>&1); F=$(find gpurun_out/t${t}_stats -name "*kernel_stats.csv" | head -1); cp $F gpurun_out/t${t}_kernel_stats.csv; rm -rf gpurun_out/t${t}_stats; done
