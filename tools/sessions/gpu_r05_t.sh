#!/bin/bash
# round 5, session t: 4:2:2 -- the partly filled second colour round served by the workgroup's last (halo) wave instead of its
# first (ZJ_ROUND_ROT=1, the product) against libzjhip_norot.so; parity of the 4:2:2 cases first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter exit $?" | tee -a $O/summary.txt
tail -2 $O/pytest.log | tee -a $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_norot.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --no-single-frame --workload 422-rgb 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib 422-rgb', d['value'], 'kernel', r['kernel_ms'], 'frac', r['frac'])" | tee -a $O/summary.txt
done; done
ZJ_RAGGED_B=60 python tools/ragged_bench.py 2500x1786 4096x4096 2>&1 | grep "422->" | tee -a $O/summary.txt
