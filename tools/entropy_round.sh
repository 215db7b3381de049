#!/bin/bash
# GPU box: the device entropy stage -- tests, per-phase timing, rocprofv3 kernel stats, file-batch rates.
# Usage: bash tools/entropy_round.sh <tag> [steps: test time prof files]
set -u
TAG=${1:-ent}
shift || true
STEPS="${*:-test time prof files}"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has test; then
  timeout 900 python -m pytest tests/test_gpu_entropy.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
fi
if has time; then
  timeout 600 python tools/entropy_gpu.py 4096 > $O/timing.txt 2>&1; cat $O/timing.txt
fi
if has prof; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o huff -- python3 $R/tools/entropy_prof.py 20 > $O/prof.log 2>&1)
  grep rounds $O/prof.log
  find $O/prof -name "*kernel_stats*.csv" | head -1 | xargs -r cat | cut -d, -f1-8 | cut -c1-160 | tee $O/kernel_stats.txt
fi
if has files; then
  timeout 600 python tools/files_bench.py --entropy gpu --files 64 > $O/files_gpu.txt 2>&1; cat $O/files_gpu.txt
fi
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -size +3M -delete
