#!/bin/bash
# round 2 diagnostics: PMC counters and ablation for the three kernel variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r02b}; mkdir -p $O; cd $R; export TMPDIR=/tmp
echo "== torch sanity" | tee $O/summary.txt
python -c "import torch; print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())" 2>&1 | tail -1 | tee -a $O/summary.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decode_to_tensor" 2>&1 | tail -3 | tee -a $O/summary.txt
for v in wide packed packed-direct; do
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $line --output-format csv -d $O/$v/p$i -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --variant $v > $O/$v-p$i.log 2>&1)
done <<LIST
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum
LIST
echo "== $v" | tee -a $O/summary.txt
python tools/pmc_summary.py $O/$v --tag $v 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['counters'].items()): print(f'{k:44s} {v[\"mean\"]:16.1f}')
" | tee -a $O/summary.txt
done
echo "== ablation" | tee -a $O/summary.txt
for v in 1 0 2; do echo "variant $v" | tee -a $O/summary.txt; ZJ_VARIANT=$v ZJ_LIB=libzjhip_ablate.so timeout 300 python tools/ablate.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; done
find $O -name "*.csv" -size +2M -delete
