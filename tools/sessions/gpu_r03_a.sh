#!/bin/bash
# Round 3, GPU session A: parity suite on the new build (hidden visibility, lab split), the new bench line (self-launch,
# shard walk, golden checksums, live traffic), a 2-rank same-GPU artefact, the LDS-DMA read lab, the occupancy probe
# for one-frame launches, rocprofv3 kernel stats.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R; export TMPDIR=/tmp
S=$O/summary.txt; : > $S
echo "== pytest -m gpu" | tee -a $S
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $S; tail -6 $O/pytest.log | tee -a $S
echo "== smoke" | tee -a $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $S
echo "== bench (default)" | tee -a $S
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $S; tail -c 3000 $O/bench.json | tee -a $S; tail -3 $O/bench.err | tee -a $S
echo "== bench --legacy-data (rounds 1-2 input)" | tee -a $S
timeout 600 python bench.py --legacy-data --no-cpu-baseline --no-live-traffic > $O/bench_legacy.json 2> $O/bench_legacy.err; tail -c 1500 $O/bench_legacy.json | tee -a $S
echo "== bench --gpus 2, both ranks on cuda:0 (ZJ_BENCH_SAME_GPU=1)" | tee -a $S
ZJ_BENCH_SAME_GPU=1 timeout 900 python bench.py --gpus 2 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err; echo "exit $?" | tee -a $S; tail -c 2500 $O/bench_2rank_same_gpu.json | tee -a $S; tail -3 $O/bench_2rank.err | tee -a $S
echo "== lab (memory patterns incl. LDS-DMA reads)" | tee -a $S
timeout 300 python tools/lab.py > $O/lab.txt 2>&1; tail -40 $O/lab.txt | tee -a $S
echo "== occupancy probe (diagnostic build), 16 frames and 1 frame per launch" | tee -a $S
ZJ_LIB=libzjhip_ablate.so timeout 600 python tools/occupancy.py > $O/occupancy.txt 2>&1; cat $O/occupancy.txt | grep -v amdgpu.ids | tee -a $S
echo "== rocprofv3 --kernel-trace --stats" | tee -a $S
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-single-frame --no-live-traffic > $O/prof_stats.log 2>&1)
grep -h '"metric"' $O/prof_stats.log | tail -1 | cut -c1-400 | tee -a $S
find $O/prof_stats -name "*kernel_stats*.csv" | head -1 | xargs -r head -8 | tee -a $S
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -size +3M -delete
du -sh $O | tee -a $S
