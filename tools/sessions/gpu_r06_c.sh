#!/bin/bash
# round 6, session c: the whole GPU suite, the default bench line, the NUMA placement A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -x -m gpu -n 4 > $O/gputest.txt 2>&1; echo "pytest rc $?"; tail -5 $O/gputest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 1500 python tools/numa_ab.py > $O/numa_ab.txt 2>&1; echo "numa rc $?"; cat $O/numa_ab.txt
