#!/bin/bash
# round 5, session n: rocprofv3 --kernel-trace --stats of the bench command with only the shard's launches in it; the pixel
# soak over every entry point (packed, host-scattered, device-scattered, strided, two-slot multi) for ten minutes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R; export TMPDIR=/tmp
bash tools/gpu_round.sh r05n/round prof > $O/round.log 2>&1; tail -8 $O/round.log | cut -c1-200 | tee $O/summary.txt
timeout 900 python tools/pixel_soak.py --seconds 600 --seed 55 > $O/pixel_soak.txt 2>&1; echo "soak exit $?" | tee -a $O/summary.txt
tail -30 $O/pixel_soak.txt | tee -a $O/summary.txt
