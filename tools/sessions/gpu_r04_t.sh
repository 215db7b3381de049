#!/bin/bash
# round 4, session t: colour rounds with the next round's LDS inputs prefetched (ZJ_COLOR_PF=1): parity, A/B
O=gpurun_out/r04t; mkdir -p $O
ZJ_LIB=libzjhip_pf1.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $O/summary.txt
timeout 1200 bash tools/ab_libs.sh libzjhip.so libzjhip_pf1.so libzjhip.so libzjhip_pf1.so libzjhip.so libzjhip_pf1.so 2>&1 | tee $O/ab.txt
for wl in 422-rgb 420-rgba; do for lib in libzjhip.so libzjhip_pf1.so libzjhip.so libzjhip_pf1.so; do
  ZJ_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-live-traffic --no-e2e --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl $lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'])"
done; done 2>&1 | tee $O/workloads.txt
