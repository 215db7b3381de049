#!/bin/bash
# round 4, session d: instruction ledger (cut points), C++ strip-range split, A/B base vs current, workloads table,
# the full bench line (e2e_pinned, per_rank_ms, ...) and --gather-rgb at N = 1
O=gpurun_out/r04d; mkdir -p $O
bash tools/valu_ledger.sh r04d/ledger > $O/ledger.log 2>&1; cat $O/ledger/ledger_counters.txt
echo "== strip-range split inside the library (diagnostic build, ZJ_SPLIT)" | tee $O/split_cpp.txt
for sp in 0 2 4; do
  echo "-- ZJ_SPLIT=$sp" | tee -a $O/split_cpp.txt
  ZJ_LIB=libzjhip_ablate.so ZJ_SPLIT=$sp timeout 300 python tools/single_frame_ab.py 0 16 2>&1 | grep ZJ_STAGGER | tee -a $O/split_cpp.txt
done
timeout 900 bash tools/ab_libs.sh libzjhip_base.so libzjhip.so libzjhip_base.so libzjhip.so 2>&1 | tee $O/ab.txt
timeout 900 bash tools/workloads.sh 2>&1 | tee $O/workloads.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 4000 $O/bench.json; tail -3 $O/bench.err
timeout 600 python bench.py --gather-rgb --shard-frames 32 --no-cpu-baseline --no-live-traffic --no-e2e > $O/bench_gather.json 2> $O/bench_gather.err; tail -c 1500 $O/bench_gather.json
