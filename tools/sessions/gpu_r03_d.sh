#!/bin/bash
# Round 3, session D: LDS bank conflicts by phase (ablation masks under rocprofv3 --pmc)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R; export TMPDIR=/tmp ZJ_LIB=libzjhip_ablate.so
S=$O/summary.txt; : > $S
for mask in 0 32 16 2 1; do
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/m$mask -o pmc -- python3 $R/tools/lds_phase.py $mask > $O/m$mask.log 2>&1)
  grep "^mask" $O/m$mask.log | tee -a $S
  python tools/pmc_summary.py $O/m$mask --tag m$mask 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('   '+'  '.join(f'{k}={v[\"mean\"]/1e6:.2f}M' for k,v in sorted(d['counters'].items())))" | tee -a $S
done
find $O -name "*.csv" -size +2M -delete
