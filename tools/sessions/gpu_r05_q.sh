#!/bin/bash
# round 5, session q: the multi-device pool (two slots on one GPU) under random batches for five minutes per entropy setting,
# every file against its own CPU-walker decode; host and device outputs, the batch entry point
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for e in gpu cpu; do
  timeout 600 python tools/pool_soak.py --seconds 240 --seed 7 --devices 0,0 --entropy $e > $O/pool_soak_$e.txt 2>&1; echo "pool soak ($e entropy, devices 0,0) exit $?" | tee -a $O/summary.txt
  tail -3 $O/pool_soak_$e.txt | tee -a $O/summary.txt
done
