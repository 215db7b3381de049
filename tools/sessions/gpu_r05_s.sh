#!/bin/bash
# round 5, session s: first-wave stagger delay once more, now that the prologue is shorter (pin_head): back-to-back and
# isolated one-frame launches under ZJ_STAGGER = 0 8 12 16 20 24 32
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R; export TMPDIR=/tmp
python tools/single_frame_ab.py 0 8 12 16 20 24 32 2>&1 | grep -v amdgpu.ids | tee $O/summary.txt
