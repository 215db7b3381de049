#!/bin/bash
# round 6, session l: decode_buffer on 1 / 4 / 7 threads against one thread (streamed and not), the pool's short batches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bench.py -q -x -m gpu > $O/stream_tests.txt 2>&1; echo "pytest rc $?"; tail -5 $O/stream_tests.txt
timeout 900 python tools/pool_short_batch.py > $O/pool_short_batch.txt 2>&1; echo "short batch rc $?"; cat $O/pool_short_batch.txt
