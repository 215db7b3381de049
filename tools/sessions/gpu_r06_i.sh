#!/bin/bash
# round 6, session i: scans without restart markers on several threads (scan_baseline_parallel) on the GPU host's EPYC
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
for t in 1 2 4 8 16; do timeout 300 python tools/walker_bench.py --pinned --no-pillow --threads $t --reps 9 2>&1 | tail -5 | sed "s/^/threads $t: /"; done > $O/walker_threads.txt 2>&1; cat $O/walker_threads.txt
timeout 300 python tools/par_scan_soak.py --seconds 120 --seed 7 > $O/par_scan_soak.txt 2>&1; tail -2 $O/par_scan_soak.txt
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_ref_images.py tests/test_gpu_bench.py -q -x -m gpu -k "not virtual_rank" > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -4 $O/tests.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
