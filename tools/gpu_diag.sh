#!/bin/bash
# diagnostics for one kernel variant: SQ / LDS / TCP counters + ablation table.  usage: bash tools/gpu_diag.sh <tag> <variant-name> <variant-number>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-diag}; V=${2:-packed}; VN=${3:-0}; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $line --output-format csv -d $O/$V/p$i -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --child --shard-frames 16 --variant $V > $O/$V-p$i.log 2>&1)
done <<LIST
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
LIST
echo "== $V" | tee -a $O/summary.txt
python tools/pmc_summary.py $O/$V --tag $V 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['counters'].items()): print(f'{k:44s} {v[\"mean\"]:16.1f}')
" | tee -a $O/summary.txt
echo "== ablation variant $VN" | tee -a $O/summary.txt
ZJ_VARIANT=$VN ZJ_LIB=libzjhip_ablate.so timeout 300 python tools/ablate.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
find $O -name "*.csv" -size +2M -delete
