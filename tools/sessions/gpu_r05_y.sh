#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R; export TMPDIR=/tmp
for rep in 1 2 3; do
python bench.py --no-cpu-baseline --no-live-traffic --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; sc=r['scattered_batch']; print('kernel', r['kernel_ms'], r['frac'], 'scattered', sc['kernel_ms'], sc['vs_kernel_ms'], 'adjacent', sc['adjacent_frames_kernel_ms'], sc['adjacent_frames_vs_kernel_ms'], sc['same_checksums_as_contiguous'])" | tee -a $O/summary.txt
done
