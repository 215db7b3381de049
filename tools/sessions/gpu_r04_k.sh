#!/bin/bash
# round 4, session k: sparse form of IDCT pass 1 as uniform branches inside one transform body: parity, A/B, dense control
O=gpurun_out/r04k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py tests/test_ref_images.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 1200 bash tools/ab_libs.sh libzjhip_rot0.so libzjhip_nosparse.so libzjhip.so libzjhip_rot0.so libzjhip_nosparse.so libzjhip.so 2>&1 | tee $O/ab.txt
for wl in 444-gray 422-rgb 420-chw 444-rgb; do for lib in libzjhip_rot0.so libzjhip.so libzjhip_rot0.so libzjhip.so; do
  ZJ_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-live-traffic --no-e2e --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl $lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'dense', (r.get('dense_control') or {}).get('kernel_ms'))"
done; done 2>&1 | tee $O/workloads.txt
