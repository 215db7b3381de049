#!/bin/bash
# round 5, session j: rows whose pitch is not a multiple of 128 bytes (2512 px: 7536 B) cost 14 % although every store is
# 16-byte aligned -- neighbouring tiles share cache lines at their seams.  Do ordinary (cacheable) stores merge in L2 where
# the streaming ones do not?  libzjhip.so (ZJ_NT=4: non-temporal staged stores) vs libzjhip_nt0.so (ZJ_NT=0)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for lib in libzjhip.so libzjhip_nt0.so libzjhip.so libzjhip_nt0.so; do
  echo "== $lib" | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_B=60 python tools/ragged_bench.py 2560x1792 2512x1792 2500x1786 4090x4096 4096x4096 2>&1 | grep -v amdgpu.ids | grep "420->" | tee -a $O/summary.txt
done
