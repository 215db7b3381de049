#!/bin/bash
# round 5, session c: A/B again after the frame pointers moved into TileId (scalar select, no indexed kernarg load);
# bench tests; the size-limit tests; then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
grep -E "MemTotal|MemAvailable" /proc/meminfo | tee -a $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noscat.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['kernel_ms_single_launch'], s['frac'], '4 streams', s['four_streams_ms_per_frame'], '| dense', (r.get('dense_control') or {}).get('kernel_ms'))" | tee -a $O/summary.txt
done; done
python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('full', d['value'], r['kernel_ms'], r['frac'], 'valu', r['valu_issue'], 'scattered', r['scattered_batch'])" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_gpu_limits.py -m gpu -q > $O/pytest_bench_limits.log 2>&1; echo "bench+limits tests exit $?" | tee -a $O/summary.txt
tail -40 $O/pytest_bench_limits.log | tee -a $O/summary.txt
timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench.py --deselect tests/test_gpu_limits.py > $O/pytest_all.log 2>&1; echo "suite exit $?" | tee -a $O/summary.txt
tail -8 $O/pytest_all.log | tee -a $O/summary.txt
