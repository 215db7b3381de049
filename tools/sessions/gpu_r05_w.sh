#!/bin/bash
# round 5, session w: final build (no extra readfirstlane where no round rotation applies): GPU suite, bench lines, 4:2:2
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05w; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
( time timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1 ) 2>&1 | grep real | tee -a $O/summary.txt
tail -2 $O/pytest_all.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
( time ZJ_BENCH_SAME_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err ) 2>&1 | grep real | tee -a $O/summary.txt
python tools/virtual_ranks.py 2>&1 | grep -v amdgpu.ids > $O/virtual_ranks.txt; tail -1 $O/virtual_ranks.txt | tee -a $O/summary.txt
bash tools/gpu_round.sh r05w/round prof pmc sq > $O/round.log 2>&1; tail -22 $O/round.log | cut -c1-160 | tee -a $O/summary.txt
bash tools/workloads.sh > /dev/null 2>&1; cp gpurun_out/workloads.txt $O/workloads.txt; cat $O/workloads.txt | cut -c1-170 | tee -a $O/summary.txt
