#!/bin/bash
# round 4, session i: rocprofv3 stats + PMC passes of the final kernel (clean: no e2e launches in them), two RCCL ranks on one GPU
O=gpurun_out/r04i; mkdir -p $O
bash tools/gpu_round.sh r04i/round bench prof pmc sq > $O/round.log 2>&1; tail -40 gpurun_out/r04i/round/summary.txt
timeout 400 python tools/nccl_two_ranks_one_gpu.py 2>&1 | tee $O/two_ranks_one_gpu.txt
