#!/bin/bash
# round 5, session ah: the ragged family at 5 workgroups per CU without spills (93 VGPRs) against 6 with 80 bytes of scratch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05ah; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_rag5.so; do for pitch in 0 128; do
  echo "== $lib pitch $pitch" | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_PITCH=$pitch ZJ_RAGGED_B=60 python tools/ragged_bench.py 2500x1786 1366x768 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_PITCH=$pitch python tools/ragged_bench.py 4090x4096 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
done; done; done
