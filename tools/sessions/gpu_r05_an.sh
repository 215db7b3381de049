#!/bin/bash
# round 5, session an: what tools/store_probe's second table predicts -- non-temporal coefficient loads (-DZJ_NT=6) on the
# aligned headline, and with them the shared seam lines written back (-DZJ_SEAM_WB=1: 4:2:0; =2: every mode) on odd pitches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05an; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for lib in libzjhip_nt6.so libzjhip_nt6seam.so libzjhip_nt6seamall.so; do
  ZJ_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_$lib.log 2>&1; echo "$lib parity exit $?" | tee -a $O/summary.txt
done
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_nt6.so libzjhip_nt6seam.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --no-single-frame --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'])" | tee -a $O/summary.txt
done; done
for rep in 1 2; do for lib in libzjhip.so libzjhip_nt6.so libzjhip_seam.so libzjhip_nt6seam.so libzjhip_nt6seamall.so; do
  echo "== $lib" | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_B=60 python tools/ragged_bench.py 2512x1792 2500x1786 1280x720 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
  ZJ_LIB=$lib python tools/ragged_bench.py 4080x4096 4096x4096 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
done; done
