#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel stats of the default bench command,
# plus the two PMC passes (FETCH_SIZE / WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes).
# Outputs land in gpurun_out/<tag>_*; tools/summarize_profiles.py copies the summaries into profiles/.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > $ROOT/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side > $ROOT/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_write -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side > $ROOT/gpurun_out/${TAG}_write.log 2>&1
python $ROOT/bench.py --steps 5 --warmup 2 > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
echo collected $TAG
