#!/bin/bash
# round 6, session m: pass B in 4 T ordered parts, rows streamed to the GPU from inside the region
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_stream.py -q -x -m gpu > $O/stream_tests.txt 2>&1; echo "pytest rc $?"; tail -3 $O/stream_tests.txt
bash tools/sessions/gpu_r06_j.sh > $O/walker_threads.txt 2>&1; grep -v "scan_baseline" $O/walker_threads.txt | grep pillow
for t in 1 4 8; do timeout 600 python tools/stream_sweep.py --threads $t >> $O/stream.txt 2>&1; done; cat $O/stream.txt
