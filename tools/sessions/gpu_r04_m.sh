#!/bin/bash
# round 4, session m: 4:2:2 on 384-pixel tiles (ZJ_TWC_H=24) against the 256-pixel default
O=gpurun_out/r04m; mkdir -p $O
for lib in libzjhip.so libzjhip_h24.so libzjhip.so libzjhip_h24.so libzjhip.so libzjhip_h24.so; do
  ZJ_LIB=$lib timeout 300 python bench.py --workload 422-rgb --no-cpu-baseline --no-live-traffic --no-e2e --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('422-rgb $lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden/dense', (r.get('dense_control') or {}).get('kernel_ms'))"
done 2>&1 | tee $O/h24.txt
ZJ_LIB=libzjhip_h24.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "h or 422 or 2_1 or horizontal" 2>&1 | tail -2 | tee -a $O/h24.txt
