#!/bin/bash
# quick A/B on the GPU box: parity of the listed test files (optional), then the three kernel variants, twice
# usage: bash tools/gpu_ab.sh <tag> [notest]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
if [ "${2:-}" != "notest" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity exit $?" | tee -a $O/summary.txt; tail -4 $O/pytest_parity.log | tee -a $O/summary.txt
fi
for rep in 1 2; do for v in wide packed packed-direct; do
  python bench.py --no-cpu-baseline --variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('420-rgb $v', d['value'], 'MP/s', r['kernel_ms'], 'ms/launch', r['achieved'], 'GB/s', r['frac'])" | tee -a $O/summary.txt
done; done
