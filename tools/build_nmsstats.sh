#!/bin/bash
# tools/libuda_nmsstats.so = the library with kernels_post.hip compiled -DUDA_NMS_STATS (phase times, list sizes, winners per
# step, slowest block per step of nms_coop_kernel on stderr under UDA_NMS_DEBUG=1); loaded through UDA_LIB (tools/nms_stats_run.sh).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/uncertainty-detection-autolabeling_amd/csrc
make -C $C -j4 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DUDA_NMS_STATS -c $C/kernels_post.hip -o /tmp/kernels_post_stats.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/libuda_nmsstats.so $C/_build/uda_api.o $C/_build/kernels_conv.o $C/_build/kernels_pwb.o $C/_build/kernels_sep.o /tmp/kernels_post_stats.o
echo built $ROOT/tools/libuda_nmsstats.so
