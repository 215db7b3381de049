#!/bin/bash
# round 5, session p: placement of a step's planes and pixels in HBM
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R; export TMPDIR=/tmp
python tools/layout_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/summary.txt
