#!/bin/bash
# round 6, session a: the host feeder A/B and the un-skipped ragged parity cases
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
lscpu > $O/lscpu.txt 2>&1
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpuset.mems.effective > $O/cgroup.txt 2>&1
timeout 900 python tools/feeder_ab.py > $O/feeder_ab.txt 2>&1; echo "feeder rc $?"
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -k "ragged or medium_image" > $O/ragged.txt 2>&1; echo "pytest rc $?"
tail -3 $O/ragged.txt
