#!/bin/bash
# One GPU-box session: parity tests, micro-benchmark, bench, rocprofv3 stats + PMC passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag> [steps: test ubench bench prof pmc sq]
set -u
TAG=${1:-r01}
shift || true
STEPS="${*:-test smoke ubench bench prof pmc sq}"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
has() { [[ " $STEPS " == *" $1 "* ]]; }
echo "== $TAG: $STEPS" | tee $O/summary.txt
if has test; then
  echo "== pytest -m gpu" | tee -a $O/summary.txt
  timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/summary.txt
  tail -15 $O/pytest.log | tee -a $O/summary.txt
fi
if has smoke; then
  echo "== smoke" | tee -a $O/summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee -a $O/summary.txt
fi
if has ubench; then
  echo "== ubench" | tee -a $O/summary.txt
  timeout 300 python tools/ubench.py > $O/ubench.txt 2>&1; cat $O/ubench.txt | tee -a $O/summary.txt
fi
if has lab; then
  echo "== lab" | tee -a $O/summary.txt
  timeout 300 python tools/lab.py > $O/lab.txt 2>&1; cat $O/lab.txt | tee -a $O/summary.txt
fi
if has bench; then
  echo "== bench" | tee -a $O/summary.txt
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.json | tee -a $O/summary.txt; tail -3 $O/bench.err
fi
if has prof; then
  echo "== rocprofv3 --kernel-trace --stats (same command as bench)" | tee -a $O/summary.txt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-single-frame --no-live-traffic --no-e2e --no-dense-control --no-other-workloads > $O/prof_stats.log 2>&1)
  grep -h '"metric"' $O/prof_stats.log | tail -1 | tee -a $O/summary.txt
  find $O/prof_stats -name "*kernel_stats*.csv" | head -1 | xargs -r head -6 | tee -a $O/summary.txt
fi
if has pmc; then
  echo "== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)" | tee -a $O/summary.txt
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --min-untimed 5 --shard-frames 16 --child > $O/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --min-untimed 5 --shard-frames 16 --child > $O/pmc_write.log 2>&1)
  python tools/pmc_summary.py $O --out $O/pmc_summary.json --tag $TAG 2>&1 | tail -30 | tee -a $O/summary.txt
fi
if has sq; then
  echo "== rocprofv3 --pmc SQ counters" | tee -a $O/summary.txt
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --min-untimed 5 --shard-frames 16 --child > $O/pmc_sq.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq2 -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --min-untimed 5 --shard-frames 16 --child > $O/pmc_sq2.log 2>&1)
  python tools/pmc_summary.py $O/pmc_sq --tag $TAG-sq 2>&1 | grep -E '"(SQ|GRBM)|mean' | paste - - | tee -a $O/summary.txt
  python tools/pmc_summary.py $O/pmc_sq2 --tag $TAG-sq2 2>&1 | grep -E '"(SQ|GRBM)|mean' | paste - - | tee -a $O/summary.txt
fi
if has pmc && has sq; then python tools/pmc_summary.py $O --out $O/pmc_summary.json --tag $TAG > /dev/null 2>&1; fi
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -size +3M -delete
du -sh $O | tee -a $O/summary.txt
