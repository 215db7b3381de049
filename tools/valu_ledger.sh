#!/bin/bash
# Dynamic instruction ledger of the fused 4:2:0 -> RGB kernel (GPU box): the diagnostic build (-DZJ_ABLATION) ends the
# kernel at successive cut points; SQ_INSTS_VALU / SALU / LDS / VMEM of two cuts differ by exactly one phase.
# usage: bash tools/valu_ledger.sh <tag>      needs zune-jpeg_amd/libzjhip_ablate.so (tools/build_variant.sh ablate -DZJ_ABLATION=1)
TAG=${1:-ledger}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
export ZJ_LIB=libzjhip_ablate.so
for m in 32 64 128 256 0 1 2 16; do
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $O/m$m -o pmc -- python3 $R/tools/ledger_run.py $m > $O/m$m.log 2>&1)
done
cd $R
python3 - "$O" <<'PY' | tee $O/ledger_counters.txt
import json, subprocess, sys
O = sys.argv[1]
names = {32: "cut A: after tile decode, block addresses, load issue, table staging", 64: "cut B: + classification, IDCT, LDS staging / halo wave",
         128: "cut C: + colour round 0: luma unpack, filters, colour math, packing", 256: "cut D: + round 0 staging and stores",
         0: "full kernel", 1: "full, IDCT skipped (all blocks take the DC-only path)", 2: "full, colour math skipped", 16: "full, chroma LDS reads and filters skipped"}
rows = {}
for m in names:
    out = subprocess.run([sys.executable, "tools/pmc_summary.py", f"{O}/m{m}", "--tag", f"m{m}"], capture_output=True, text=True).stdout
    try:
        c = json.loads(out)["counters"]
    except Exception:
        continue
    rows[m] = {k: v["mean"] for k, v in c.items()}
tiles = 16 * 2048
print(f"{'mask':>5s} {'VALU':>12s} {'SALU':>12s} {'LDS':>11s} {'VMEM_RD':>10s} {'VMEM_WR':>10s} | per tile: VALU  SALU   LDS   what")
for m in (32, 64, 128, 256, 0, 1, 2, 16):
    r = rows.get(m)
    if not r: continue
    g = lambda k: r.get(k, float('nan'))
    print(f"{m:5d} {g('SQ_INSTS_VALU'):12.0f} {g('SQ_INSTS_SALU'):12.0f} {g('SQ_INSTS_LDS'):11.0f} {g('SQ_INSTS_VMEM_RD'):10.0f} {g('SQ_INSTS_VMEM_WR'):10.0f} |"
          f" {g('SQ_INSTS_VALU') / tiles:9.1f} {g('SQ_INSTS_SALU') / tiles:6.1f} {g('SQ_INSTS_LDS') / tiles:6.1f}   {names[m]}")
json.dump(rows, open(f"{O}/ledger_counters.json", "w"), indent=1)
PY
find $O -name "*.csv" -size +2M -delete
