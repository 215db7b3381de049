#!/bin/bash
# round 4, session f: wave-role rotation A/B, parity on the rotated kernels, workloads
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 900 bash tools/ab_libs.sh libzjhip_rot0.so libzjhip.so libzjhip_rot0.so libzjhip.so libzjhip_base.so 2>&1 | tee $O/ab.txt
for wl in 444-rgb 444-gray 422-rgb 440-rgb 420-rgba 420-chw; do for lib in libzjhip_rot0.so libzjhip.so libzjhip_rot0.so libzjhip.so; do
  ZJ_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-live-traffic --no-e2e --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl $lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'])"
done; done 2>&1 | tee $O/workloads.txt
