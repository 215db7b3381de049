#!/bin/bash
# round 5, session a: the new entry points' GPU parity tests, then the unchanged headline (did the pointer table or the
# stagger gating move the hot kernel?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scatter.py -m gpu -x -q > $O/pytest_scatter.log 2>&1; echo "scatter exit $?" | tee $O/summary.txt
tail -15 $O/pytest_scatter.log | tee -a $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-e2e > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
tail -c 3000 $O/bench.json | tee -a $O/summary.txt; tail -3 $O/bench.err | tee -a $O/summary.txt
