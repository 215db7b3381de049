#!/bin/bash
# round 5, session b: (1) A/B on one box: the library with the pointer table in the kernel arguments against a build of the
# same sources without it (/tmp/ab build, libzjhip_noscat.so) -- did the table cost the contiguous launches anything?
# (2) bench.py's new fields through tests/test_gpu_bench.py; (3) the full default line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noscat.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['kernel_ms_single_launch'], s['frac'], '4 streams', s['four_streams_ms_per_frame'], '| dense', (r.get('dense_control') or {}).get('kernel_ms'))" | tee -a $O/summary.txt
done; done
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -x -q > $O/pytest_bench.log 2>&1; echo "bench tests exit $?" | tee -a $O/summary.txt
tail -25 $O/pytest_bench.log | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
tail -c 6000 $O/bench.json | tee -a $O/summary.txt; tail -3 $O/bench.err | tee -a $O/summary.txt
