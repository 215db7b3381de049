#!/bin/bash
# round 5, session ao: the seam kernel family (4:2:0, aligned widths, rows off the 128-byte grid: shared lines written back)
# as the default -- parity, then odd-pitch frames and the aligned headline against -DZJ_SEAM_WB=0
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05ao; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "suite exit $?" | tee -a $O/summary.txt
tail -2 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noseam.so; do
  echo "== $lib" | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_MODES=420 ZJ_RAGGED_B=60 python tools/ragged_bench.py 2512x1792 1840x1040 1600x1200 720x480 2560x1792 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-135 | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_MODES=420 python tools/ragged_bench.py 4080x4096 4096x4096 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-135 | tee -a $O/summary.txt
done; done
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noseam.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['frac'])" | tee -a $O/summary.txt
done; done
