#!/bin/bash
# parity (all GPU tests, with durations) + workloads table.  usage: bash tools/gpu_check.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-chk}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest.log 2>&1; echo "pytest exit $?" | tee $O/summary.txt; tail -14 $O/pytest.log | tee -a $O/summary.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/summary.txt
bash tools/workloads.sh $O/workloads.txt 2>&1 | tee -a $O/summary.txt
