#!/bin/bash
# round 4, session g: wave-role rotation with the thread-number range restored for the compiler, A/B
O=gpurun_out/r04g; mkdir -p $O
timeout 900 bash tools/ab_libs.sh libzjhip_rot0.so libzjhip.so libzjhip_rot0.so libzjhip.so 2>&1 | tee $O/ab.txt
for wl in 444-rgb 422-rgb 420-rgba; do for lib in libzjhip_rot0.so libzjhip.so libzjhip_rot0.so libzjhip.so; do
  ZJ_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-live-traffic --no-e2e --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl $lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'])"
done; done 2>&1 | tee $O/workloads.txt
