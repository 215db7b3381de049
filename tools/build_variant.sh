#!/bin/bash
# build_variant.sh NAME "EXTRA FLAGS": zune-jpeg_amd/libzjhip_NAME.so with zj_kernels.hip and zj_api.cpp recompiled under
# EXTRA (the other objects are reused); select it at run time with ZJ_LIB=libzjhip_NAME.so.  A/B experiments and the
# diagnostic build (NAME = ablate, EXTRA = -DZJ_ABLATION: ablation switches, occupancy probe) only.
# Linked with the product's own flags (the Makefile's: hidden visibility, --exclude-libs,ALL) but with a soname of its
# own, so that a variant can never be picked up as the product library (the version script exports zj_*: the ablate build's
# zj_set_ablation / zj_set_pad_lds / zj_fused_occupancy are the only additions).
set -e
cd "$(dirname "$0")/../zune-jpeg_amd/csrc"
make -s
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS $2 -c zj_kernels.hip -o /tmp/zj_kernels_$1.o
# zj_api.cpp sees the tile geometry through zj_plan.h: recompile it under the same flags
/opt/rocm/bin/hipcc $FLAGS $2 -x hip -c zj_api.cpp -o /tmp/zj_api_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libzjhip_$1.so /tmp/zj_kernels_$1.o zj_huff.o /tmp/zj_api_$1.o zj_jpeg.o zj_pool.o zj_multi.o \
  -Wl,-soname,libzjhip_$1.so -Wl,--exclude-libs,ALL -Wl,--version-script=zjhip.map
echo built libzjhip_$1.so
