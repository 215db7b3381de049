#!/bin/bash
# round 2, first GPU session: parity of the new kernel generation, instruction costs, isolated IDCT, A/B of the variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r02a}; mkdir -p $O; cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu (test_gpu_parity only first)" | tee $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity exit $?" | tee -a $O/summary.txt; tail -5 $O/pytest_parity.log | tee -a $O/summary.txt
echo "== ubench" | tee -a $O/summary.txt
timeout 300 python tools/ubench.py > $O/ubench.txt 2>&1; grep -E "clock|dot2|sad|sdwa|ashr|mad_i32_i24 v,v,s|perm|PAIR" $O/ubench.txt | tee -a $O/summary.txt
echo "== lab" | tee -a $O/summary.txt
timeout 300 python tools/lab.py > $O/lab.txt 2>&1; head -12 $O/lab.txt | tee -a $O/summary.txt
echo "== variants A/B (two rounds)" | tee -a $O/summary.txt
for rep in 1 2; do for v in wide packed packed-direct; do
  python bench.py --no-cpu-baseline --variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('420-rgb $v', d['value'], 'MP/s', r['kernel_ms'], 'ms/launch', r['achieved'], 'GB/s', r['frac'])" | tee -a $O/summary.txt
done; done
