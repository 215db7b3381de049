#!/bin/bash
# round 6, session g: the front-end compiled with ROCm's clang (walker on the EPYC again), the streamed decode_buffer for
# several unit sizes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
timeout 300 python tools/walker_bench.py --pinned > $O/walker_clang.txt 2>&1; cat $O/walker_clang.txt
timeout 300 python tools/stream_sweep.py > $O/stream_sweep.txt 2>&1; cat $O/stream_sweep.txt
timeout 600 python -m pytest tests/test_ref_images.py tests/test_gpu_stream.py tests/test_gpu_entropy.py -q -x -m gpu > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -3 $O/tests.txt
