#!/bin/bash
# round 5, session l: line-aware lane mapping of the copy-out for pitches that are not a multiple of 128 B (ZJ_LINE_OCTETS=1,
# the product) against the plain mapping (libzjhip_nooct.so), aligned widths with such pitches; parity first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter exit $?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
for lib in libzjhip.so libzjhip_nooct.so libzjhip.so libzjhip_nooct.so; do
  echo "== $lib" | tee -a $O/summary.txt
  ZJ_LIB=$lib ZJ_RAGGED_B=60 python tools/ragged_bench.py 2560x1792 2512x1792 2544x1792 2528x1792 3024x4032 4032x3024 4096x4096 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
