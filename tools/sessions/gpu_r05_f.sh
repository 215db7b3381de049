#!/bin/bash
# round 5, session f: the ragged kernel family (one launch: fast stores for a row's ordinary groups, the row end from LDS by
# the generic rules) -- parity, timing; A/B of the pinned kernel-argument head (ZJ_PIN_ARGS) on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py tests/test_gpu_limits.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter+limits exit $?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_nopin.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; sc=r['scattered_batch']; print('$lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['kernel_ms_single_launch'], s['frac'], '4 streams', s['four_streams_ms_per_frame'], '| dense', (r.get('dense_control') or {}).get('kernel_ms'), '| scattered', sc['kernel_ms'], sc['vs_kernel_ms'], sc['same_checksums_as_contiguous'], '| other', {k:v.get('frac') for k,v in d['other_workloads'].items()})" | tee -a $O/summary.txt
done; done
echo "== ragged" | tee -a $O/summary.txt
python tools/ragged_bench.py 2500x1786 2512x1786 4090x4096 4095x4096 4096x4096 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
