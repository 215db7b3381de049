#!/bin/bash
# Diagnostic PMC passes for the fused kernel (LDS / TA / TCP / TCC stalls).  bash tools/pmc_diag.sh <tag> [variant]
TAG=${1:-diag}; VAR=${2:-onepass}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $line --output-format csv -d $O/p$i -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --child --shard-frames 16 --variant $VAR > $O/p$i.log 2>&1)
done <<LIST
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum
TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_sum TCC_BUSY_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY
LIST
cd $R; python tools/pmc_summary.py $O --tag $TAG 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['counters'].items()): print(f'{k:44s} {v[\"mean\"]:16.1f}')
" | tee $O/diag.txt
find $O -name "*.csv" -size +2M -delete
