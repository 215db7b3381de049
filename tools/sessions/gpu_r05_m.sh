#!/bin/bash
# round 5, session m: the round's evidence with the final build -- whole GPU suite, smoke, the bench lines (default, the
# driver's N = 1 shape, the 2-rank same-GPU dry run), virtual ranks, rocprofv3 stats of the same command, PMC passes,
# the workload table, ragged widths
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
( time timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1 ) 2>&1 | grep real | tee -a $O/summary.txt
tail -4 $O/pytest_all.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
( time ZJ_BENCH_SAME_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err ) 2>&1 | grep real | tee -a $O/summary.txt
python tools/virtual_ranks.py 2>&1 | grep -v amdgpu.ids > $O/virtual_ranks.txt; tail -2 $O/virtual_ranks.txt | tee -a $O/summary.txt
bash tools/gpu_round.sh r05m/round prof pmc sq > $O/round.log 2>&1; tail -30 $O/round.log | cut -c1-220 | tee -a $O/summary.txt
bash tools/workloads.sh > $O/workloads.txt 2>&1; cat $O/workloads.txt | cut -c1-200 | tee -a $O/summary.txt
ZJ_RAGGED_B=60 python tools/ragged_bench.py 2500x1786 2512x1786 2560x1792 4090x4096 4096x4096 2>&1 | grep -v amdgpu.ids | tee $O/ragged.txt | tee -a $O/summary.txt
