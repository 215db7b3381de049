#!/bin/bash
# One GPU-box session: parity tests, micro-benchmark, bench, rocprofv3 stats + PMC passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag> [pytest-args]
set -u
TAG=${1:-r01}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== build check" | tee $O/summary.txt
ls -la zune-jpeg_amd/libzjhip.so oracle/libzjoracle.so >> $O/summary.txt 2>&1
echo "== pytest -m gpu" | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
echo "== smoke" | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $O/summary.txt 2>&1
echo "== ubench" | tee -a $O/summary.txt
timeout 300 python tools/ubench.py > $O/ubench.txt 2>&1; cat $O/ubench.txt | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.json | tee -a $O/summary.txt; tail -3 $O/bench.err
echo "== rocprofv3 stats" | tee -a $O/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python3 $R/bench.py --steps 30 --no-cpu-baseline > $O/prof_stats.log 2>&1
find $O/prof_stats -name "*kernel_stats*.csv" | head -1 | xargs -r head -8 | tee -a $O/summary.txt
echo "== rocprofv3 pmc" | tee -a $O/summary.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
mkdir -p $O/pmc && find $O/pmc_fetch $O/pmc_write -name "*counter_collection*.csv" | while read f; do cp "$f" "$O/pmc/$(echo $f | tr '/' '_' | tail -c 80)"; done
cd $R
python tools/pmc_summary.py $O/pmc --out $O/pmc_summary.json --tag $TAG 2>&1 | tail -25 | tee -a $O/summary.txt
# keep the merged-back payload small
find $O -name "*.csv" -size +3M -delete
du -sh $O | tee -a $O/summary.txt
