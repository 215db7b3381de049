#!/bin/bash
# round 5, session h: rows that are not dword-aligned are copied out shifted (v_alignbyte): parity, ragged timings again;
# the PCIe probe with frame-sized copies; bench tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scatter.py tests/test_gpu_limits.py -m gpu -q -x > $O/pytest.log 2>&1; echo "parity+scatter+limits exit $?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
echo "== ragged" | tee -a $O/summary.txt
python tools/ragged_bench.py 2500x1786 2501x1786 2502x1786 2512x1786 4090x4096 4095x4096 4096x4096 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q > $O/pytest_bench.log 2>&1; echo "bench tests exit $?" | tee -a $O/summary.txt
tail -6 $O/pytest_bench.log | tee -a $O/summary.txt
python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('e2e', d['e2e_pinned']); print('other', {k:(v.get('frac'),v.get('kernel_ms')) for k,v in d['other_workloads'].items()})" | tee -a $O/summary.txt
