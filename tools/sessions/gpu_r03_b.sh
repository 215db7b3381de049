#!/bin/bash
# Round 3, session B: A/B of kernel builds (tools/ab_libs.sh) + quick parity of the new default build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R; export TMPDIR=/tmp
S=$O/summary.txt; : > $S
echo "== parity (new default build)" | tee -a $S
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $S; tail -4 $O/pytest.log | tee -a $S
echo "== A/B" | tee -a $S
bash tools/ab_libs.sh "$@" 2>&1 | tee -a $S
