#!/bin/bash
# round 5, session k: pitch alignment vs partial last tile.  2432 and 2688 px: pitch a multiple of 128 B, half-empty last tile;
# 2576 px: pitch NOT a multiple of 128 B, last tile holds one group; 2560: the aligned full-tile reference
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R; export TMPDIR=/tmp
ZJ_RAGGED_B=60 python tools/ragged_bench.py 2560x1792 2432x1792 2688x1792 2576x1792 2512x1792 2544x1792 2528x1792 2>&1 | grep -v amdgpu.ids | grep "420->\|444->" | tee $O/summary.txt
