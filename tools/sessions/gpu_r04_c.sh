#!/bin/bash
# round 4, session c: stagger sweep (128-cycle steps, modes), strip-range split, A/B base vs magic division (4 reps),
# 4:2:2 / 4:4:0 tile variants, new malformed-blob checks
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_entropy.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 900 python tools/single_frame_ab.py 0 12 14 16 18 20 24 272 280 528 536 784 2>&1 | grep -v Warning | tee $O/stagger.txt
timeout 600 python tools/single_frame_split.py 2>&1 | grep -v Warning | tee $O/split.txt
timeout 900 bash tools/ab_libs.sh libzjhip_base.so libzjhip.so libzjhip_base.so libzjhip.so 2>&1 | tee $O/ab.txt
for wl in 422-rgb 440-rgb; do for lib in libzjhip.so libzjhip_h256.so libzjhip_v32.so libzjhip_v32n.so libzjhip.so libzjhip_h256.so libzjhip_v32.so libzjhip_v32n.so; do
  ZJ_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-live-traffic --no-e2e --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl $lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'])"
done; done 2>&1 | tee $O/workloads.txt
