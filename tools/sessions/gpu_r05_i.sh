#!/bin/bash
# round 5, session i: why is a 2512x1786 frame (aligned) at 0.60 when 4096x4096 is at 0.70?  Sizes that isolate partial
# tiles, clipped strips and launch length; 60 frames per launch for the small sizes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
ZJ_RAGGED_B=60 python tools/ragged_bench.py 2560x1792 2512x1792 2560x1786 2304x1792 2048x2048 4096x2048 2500x1786 2512x1786 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
echo "== 32 frames per launch" | tee -a $O/summary.txt
python tools/ragged_bench.py 2560x1792 2512x1786 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
