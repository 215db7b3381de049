#!/bin/bash
# GPU box: kernel trace of tools/entropy_batch.py; prints the kernels of the last batch (16 files) with start and duration
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-batchtrace}
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/tools/entropy_batch.py > $O/log.txt 2>&1)
grep batch $O/log.txt
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last batch: from the last sync kernel with 16 jobs whose predecessor is not a sync kernel
idx = [i for i, r in enumerate(rows) if "sync" in r["Kernel_Name"] and r["Grid_Size_Y"] == "16" and (i == 0 or "sync" not in rows[i - 1]["Kernel_Name"])]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
agg = {}
for r in last:
    k = r["Kernel_Name"].split("(")[0][-28:]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(k, []).append(d)
for k, v in agg.items():
    print(f"{k:30s} x{len(v):3d}  total {sum(v):8.1f} us   each " + " ".join(f"{x:.0f}" for x in v[:12]))
print(f"span {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
PY
