#!/bin/bash
# round 6, session r: the closing run on the final build -- session k's commands (the driver's own + walker / stream / soaks), the
# parity suites on the all-variants build, the pool soak on two slots with the CPU walker (short batches lend threads to their
# files: the parallel scan inside the pool), the pixel soak, the reference's benchmark, the pool's short batches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; export ZJ_SESSION=${ZJ_SESSION:-r06r}; O=$R/gpurun_out/$ZJ_SESSION; mkdir -p $O; cd $R
bash tools/sessions/gpu_r06_k.sh
ZJ_LIB=libzjhip_all.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pitch.py tests/test_gpu_scatter.py -q -x -m gpu -n 4 > $O/gputest_all_variants.txt 2>&1; echo "pytest all-variants rc $?"; tail -3 $O/gputest_all_variants.txt
timeout 400 python tools/pool_soak.py --seconds 120 --devices 0,0 --entropy cpu > $O/pool_soak_cpu.txt 2>&1; echo "pool soak cpu rc $?"; tail -2 $O/pool_soak_cpu.txt
timeout 400 python tools/pixel_soak.py --seconds 120 > $O/pixel_soak.txt 2>&1; echo "pixel soak rc $?"; tail -3 $O/pixel_soak.txt
timeout 900 python tools/pool_short_batch.py > $O/pool_short_batch.txt 2>&1; echo "short batch rc $?"; cat $O/pool_short_batch.txt
for t in 1 4 8 16; do timeout 300 python tools/walker_bench.py --pinned --no-pillow --threads $t --reps 7 2>&1 | grep -E "^speed_bench" | sed "s/^/threads $t: /"; done | tee $O/walker_speed_bench.txt
