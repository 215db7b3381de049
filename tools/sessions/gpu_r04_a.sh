#!/bin/bash
# round 4, session a: toolchain probe (cargo), RCCL world-size-1 probe with and without HSA_ENABLE_IPC_MODE_LEGACY, baseline bench
O=gpurun_out/r04a; mkdir -p $O
{ echo "== toolchains"; which cargo rustc go javac node 2>&1; ls ~/.cargo 2>&1 | head -2; rustc --version 2>&1 | head -1; nproc; cat /sys/fs/cgroup/cpu.max; rocm-smi --showtopo 2>&1 | head -20; echo HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY; } > $O/probe.txt 2>&1
echo "== nccl probe (env as given)" >> $O/probe.txt
timeout 300 python tools/nccl_probe.py >> $O/probe.txt 2>&1; echo "rc $?" >> $O/probe.txt
echo "== nccl probe (HSA_ENABLE_IPC_MODE_LEGACY unset)" >> $O/probe.txt
( unset HSA_ENABLE_IPC_MODE_LEGACY; timeout 300 python tools/nccl_probe.py ) >> $O/probe.txt 2>&1; echo "rc $?" >> $O/probe.txt
echo "== nccl probe (HSA_ENABLE_IPC_MODE_LEGACY=1)" >> $O/probe.txt
( export HSA_ENABLE_IPC_MODE_LEGACY=1; timeout 300 python tools/nccl_probe.py ) >> $O/probe.txt 2>&1; echo "rc $?" >> $O/probe.txt
cat $O/probe.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
