#!/bin/bash
# round 4, session l: non-temporal coefficient loads A/B, then the round's evidence on the final kernel
O=gpurun_out/r04l; mkdir -p $O
timeout 900 bash tools/ab_libs.sh libzjhip.so libzjhip_nt6.so libzjhip.so libzjhip_nt6.so 2>&1 | tee $O/ab_nt.txt
bash tools/gpu_round.sh r04l/round test smoke bench prof pmc sq > $O/round.log 2>&1; tail -45 gpurun_out/r04l/round/summary.txt | cut -c1-200
bash tools/valu_ledger.sh r04l/ledger > $O/ledger.log 2>&1; cat $O/ledger/ledger_counters.txt
timeout 900 bash tools/workloads.sh 2>&1 | tee $O/workloads.txt
