#!/bin/bash
# round 6, session h: the round-end rehearsal on the closing build -- the driver's own commands
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
ls -la zune-jpeg_amd/*.so > $O/libs.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/gputest.txt 2>&1; echo "pytest rc $?"; tail -6 $O/gputest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; cat $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err; echo "driver bench rc $?"; tail -4 $O/bench_driver_n1.err
ZJ_BENCH_SAME_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err; echo "2-rank rc $?"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 600 python tools/virtual_ranks.py > $O/virtual_ranks.txt 2>&1; echo "virtual ranks rc $?"; tail -3 $O/virtual_ranks.txt
# experiment: do the pool's submitters poll away CPU quota the entropy workers could use?  (hipDeviceScheduleBlockingSync)
for mode in 0 1; do
  echo "== ZJ_BLOCKING_SYNC=$mode" >> $O/pool_sync.txt
  ZJ_BLOCKING_SYNC=$mode timeout 600 python tools/files_bench.py --files 48 --distinct 3 --restart-rows 0 2>&1 | grep -E "^zj_pool" >> $O/pool_sync.txt
done
cat $O/pool_sync.txt
