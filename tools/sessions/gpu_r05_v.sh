#!/bin/bash
# round 5, session v: the final build once more -- whole GPU suite, smoke, the bench line, the workload table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
( time timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1 ) 2>&1 | grep real | tee -a $O/summary.txt
tail -3 $O/pytest_all.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
bash tools/workloads.sh > /dev/null 2>&1; cp gpurun_out/workloads.txt $O/workloads.txt; cat $O/workloads.txt | cut -c1-170 | tee -a $O/summary.txt
