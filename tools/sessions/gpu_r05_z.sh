#!/bin/bash
# round 5, session z: the bench lines with the scattered timing warmed like the timed region (it was measured right after the
# one-frame launches, with the clocks low: the "3 % for frames that lie apart" of earlier sessions was that)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
( time ZJ_BENCH_SAME_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err ) 2>&1 | grep real | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q > $O/pytest_bench.log 2>&1; echo "bench tests exit $?" | tee -a $O/summary.txt; tail -2 $O/pytest_bench.log | tee -a $O/summary.txt
