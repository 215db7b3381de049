#!/bin/bash
# round 6, session f: after the fixes of session e -- the stream tests alone, then the whole GPU suite, the C examples, the bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_examples.py -q -x -m gpu > $O/stream.txt 2>&1; echo "stream rc $?"; tail -5 $O/stream.txt
timeout 1500 python -m pytest tests -q -m gpu -n 4 > $O/gputest.txt 2>&1; echo "pytest rc $?"; tail -5 $O/gputest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
