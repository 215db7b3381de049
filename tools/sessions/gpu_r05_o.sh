#!/bin/bash
# round 5, session o: the driver's torchrun launch shape for N = 2 on one GPU (gloo), as a test
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05o; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bench.py -m gpu -q -k "torchrun or contract or cpu_baseline" > $O/pytest.log 2>&1; echo "exit $?" | tee $O/summary.txt
tail -15 $O/pytest.log | tee -a $O/summary.txt
