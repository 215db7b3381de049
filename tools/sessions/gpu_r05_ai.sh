#!/bin/bash
# round 5, session ai: the rows the strips never reach zeroed by ONE launch per batch (was: a memset per frame) -- parity
# (every test that decodes a frame with a dropped MCU row), then frames with an odd MCU-row count: 1280x720, 1840x1040
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05ai; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "suite exit $?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2; do
  ZJ_RAGGED_B=60 python tools/ragged_bench.py 1280x720 1840x1040 1280x704 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
done
