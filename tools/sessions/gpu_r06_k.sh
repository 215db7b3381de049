#!/bin/bash
# round 6, session k: the closing build (parallel scan, crew) through the driver's own commands, the walker by thread count,
# the streamed decode_buffer with four threads, and the soaks that touch the front-end
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${ZJ_SESSION:-r06k}; mkdir -p $O; cd $R
ls -la zune-jpeg_amd/*.so > $O/libs.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gputest.txt 2>&1; echo "pytest rc $?"; tail -6 $O/gputest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; cat $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err; echo "driver bench rc $?"; tail -4 $O/bench_driver_n1.err
ZJ_BENCH_SAME_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err; echo "2-rank rc $?"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 600 python tools/virtual_ranks.py > $O/virtual_ranks.txt 2>&1; echo "virtual ranks rc $?"; tail -3 $O/virtual_ranks.txt
bash tools/sessions/gpu_r06_j.sh > $O/walker_threads.txt 2>&1; cat $O/walker_threads.txt
for t in 1 4; do timeout 600 python tools/stream_sweep.py --threads $t >> $O/stream.txt 2>&1; echo "stream rc $?"; done; cat $O/stream.txt
timeout 400 python tools/pool_soak.py --seconds 120 > $O/pool_soak.txt 2>&1; echo "pool soak rc $?"; tail -2 $O/pool_soak.txt
timeout 400 python tools/entropy_soak.py --seconds 120 > $O/entropy_soak.txt 2>&1; echo "entropy soak rc $?"; tail -2 $O/entropy_soak.txt
