#!/bin/bash
# round 5, session x: the (frozen) entropy side after the one-line history fix: device stage vs CPU walker on random files
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 700 python tools/entropy_soak.py --seconds 400 --seed 51 > $O/entropy_soak.txt 2>&1; echo "entropy soak exit $?" | tee $O/summary.txt
tail -6 $O/entropy_soak.txt | tee -a $O/summary.txt
