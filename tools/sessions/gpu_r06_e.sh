#!/bin/bash
# round 6, session e: the streamed frame (zj_frame_*, decode_buffer with pinned planes), the whole GPU suite again, the
# bench line, rocprofv3 kernel stats + PMC passes of the bench command, the pixel soak
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_stream.py -q -x -m gpu > $O/stream.txt 2>&1; echo "stream rc $?"; tail -15 $O/stream.txt
timeout 1500 python -m pytest tests -q -m gpu -n 4 > $O/gputest.txt 2>&1; echo "pytest rc $?"; tail -5 $O/gputest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
bash tools/gpu_round.sh r06e/round prof pmc sq > $O/round.log 2>&1; echo "round rc $?"; tail -25 $O/round.log
timeout 400 python tools/pixel_soak.py --seconds 200 > $O/pixel_soak.txt 2>&1; echo "pixel soak rc $?"; tail -3 $O/pixel_soak.txt
