#!/bin/bash
# round 6, session d: the slim product build (compressed code objects, no wide generation) through the whole GPU suite; the
# parity suites once more on the all-variants build (make VARIANTS=all OUT=zune-jpeg_amd/libzjhip_all.so); smoke; the bench
# line; soaks over the restructured pool (two slots on one GPU, CPU and device entropy), the pixel entry points, the device
# entropy stage
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
ls -la zune-jpeg_amd/*.so > $O/libs.txt
timeout 1500 python -m pytest tests -q -x -m gpu -n 4 > $O/gputest_product.txt 2>&1; echo "pytest product rc $?"; tail -3 $O/gputest_product.txt
ZJ_LIB=libzjhip_all.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pitch.py tests/test_gpu_scatter.py -q -x -m gpu -n 4 > $O/gputest_all_variants.txt 2>&1; echo "pytest all-variants rc $?"; tail -3 $O/gputest_all_variants.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; cat $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 400 python tools/pool_soak.py --seconds 150 --devices 0,0 --entropy cpu > $O/pool_soak_cpu.txt 2>&1; echo "pool soak cpu rc $?"; tail -3 $O/pool_soak_cpu.txt
timeout 400 python tools/pool_soak.py --seconds 150 --devices 0,0 --entropy gpu > $O/pool_soak_gpu.txt 2>&1; echo "pool soak gpu rc $?"; tail -3 $O/pool_soak_gpu.txt
timeout 400 python tools/pixel_soak.py --seconds 150 > $O/pixel_soak.txt 2>&1; echo "pixel soak rc $?"; tail -3 $O/pixel_soak.txt
timeout 400 python tools/entropy_soak.py --seconds 150 > $O/entropy_soak.txt 2>&1; echo "entropy soak rc $?"; tail -3 $O/entropy_soak.txt
