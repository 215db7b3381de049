#!/bin/bash
# round 5, session aa: __builtin_assume(tid < HALO_T0) in front of the block waves' locate (the chroma wave's path to its first
# load loses ~80 instructions) -- parity, then A/B against -DZJ_ASSUME_NO_HALO=0
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05aa; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter exit $?" | tee -a $O/summary.txt
tail -2 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2 3 4; do for lib in libzjhip.so libzjhip_noassume.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['kernel_ms_single_launch'], s['frac'], '4 streams', s['four_streams_ms_per_frame'])" | tee -a $O/summary.txt
done; done
for rep in 1 2; do for lib in libzjhip.so libzjhip_noassume.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --no-single-frame --workload 422-rgb 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib 422-rgb', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'])" | tee -a $O/summary.txt
done; done
