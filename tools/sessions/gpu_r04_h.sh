#!/bin/bash
# round 4, session h: constant role shifts A/B, then the round's evidence on the final kernel: tests, smoke, bench, rocprofv3
# stats + PMC passes, instruction ledger, workloads, one-frame launches, the two-rank same-GPU artefact
O=gpurun_out/r04h; mkdir -p $O
timeout 900 bash tools/ab_libs.sh libzjhip.so libzjhip_shift1.so libzjhip_shift2.so libzjhip_shift3.so libzjhip.so libzjhip_shift1.so libzjhip_shift2.so libzjhip_shift3.so 2>&1 | tee $O/ab_shift.txt
bash tools/gpu_round.sh r04h/round test smoke bench prof pmc sq > $O/round.log 2>&1; cat gpurun_out/r04h/round/summary.txt | tail -60
bash tools/valu_ledger.sh r04h/ledger > $O/ledger.log 2>&1; cat $O/ledger/ledger_counters.txt
timeout 900 bash tools/workloads.sh 2>&1 | tee $O/workloads.txt
timeout 300 python tools/single_frame_ab.py 0 16 2>&1 | grep ZJ_STAGGER | tee $O/single_frame.txt
ZJ_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --shard-frames 64 --no-live-traffic > $O/bench_2rank.json 2> $O/bench_2rank.err; tail -c 1500 $O/bench_2rank.json
