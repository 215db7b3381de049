#!/bin/bash
# round 5, session ae: seam lines written back, 4:2:0 and dword-aligned rows only -- odd pitches and the aligned headline
# against -DZJ_SEAM_WB=0 (libzjhip_noseam.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05ae; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter exit $?" | tee -a $O/summary.txt
tail -2 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noseam.so; do
  echo "== $lib" | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_B=60 python tools/ragged_bench.py 2512x1792 2500x1786 1840x1040 1366x768 720x480 2>&1 | grep -v amdgpu.ids | grep "420->RGB" | cut -c1-130 | tee -a $O/summary.txt
  ZJ_LIB=$lib python tools/ragged_bench.py 4080x4096 4090x4096 4092x4096 2>&1 | grep -v amdgpu.ids | grep "420->RGB" | cut -c1-130 | tee -a $O/summary.txt
done; done
for rep in 1 2 3 4 5; do for lib in libzjhip.so libzjhip_noseam.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'])" | tee -a $O/summary.txt
done; done
