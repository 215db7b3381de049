#!/bin/bash
# round 4, session b: new tests (RCCL world size 1, hung rank, bench fields), parity of the magic-division kernels,
# A/B base (round 3 kernel) vs magic division, first-wave stagger sweep for one-frame launches
O=gpurun_out/r04b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_bench.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
timeout 600 bash tools/ab_libs.sh libzjhip_base.so libzjhip.so 2>&1 | tee $O/ab.txt
timeout 600 python tools/single_frame_ab.py 0 1 2 3 4 6 8 12 2>&1 | grep -v Warning | tee $O/stagger.txt
