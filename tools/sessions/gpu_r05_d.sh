#!/bin/bash
# round 5, session d: A/B once more -- the table's entries are global-address-space pointers now (no flat_* instructions);
# ragged widths run as fast interior + generic edge launches: parity on the GPU, then 2500x1786 timings
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noscat.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['kernel_ms_single_launch'], s['frac'], '4 streams', s['four_streams_ms_per_frame'], '| dense', (r.get('dense_control') or {}).get('kernel_ms'))" | tee -a $O/summary.txt
done; done
python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('full', d['value'], r['kernel_ms'], r['frac'], 'valu', r['valu_issue'], 'scattered', r['scattered_batch'], 'other', {k:(v.get('frac'),v.get('kernel_ms')) for k,v in d['other_workloads'].items()})" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py tests/test_gpu_limits.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter+limits exit $?" | tee -a $O/summary.txt
tail -8 $O/pytest.log | tee -a $O/summary.txt
python tools/ragged_bench.py 2>&1 | tee -a $O/summary.txt
