#!/bin/bash
# round 5, session aj: where the ragged family's 7 % go -- instruction counts of the aligned kernel on 2512 x 1786 against the
# ragged kernel on 2500 x 1786 (both with rows at a pitch of 7552 bytes), 4:2:0 -> RGB, 60 frames per launch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05aj; mkdir -p $O; cd $R; export TMPDIR=/tmp
export ZJ_RAGGED_PITCH=128 ZJ_RAGGED_B=60 ZJ_RAGGED_MODES=420
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc -o pmc -- python3 $R/tools/ragged_bench.py 2512x1786 2500x1786 > $O/pmc.log 2>&1)
grep "RGB " $O/pmc.log | cut -c1-130 | tee $O/summary.txt
python3 - $O/pmc <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "zj_fused" in k:
            acc[(k[:60], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(k[0], "grid", k[2], k[1], "launches", len(v), "mean", round(sum(v) / len(v), 1))
PY
find $O -name "*.csv" -size +2M -delete
