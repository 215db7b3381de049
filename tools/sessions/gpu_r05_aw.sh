#!/bin/bash
# round 5, session aw: the shader clock DURING the launch -- GRBM_GUI_ACTIVE (summed over the 8 XCDs) over each dispatch's own
# duration, same pass (rocprofv3 --pmc with --kernel-trace)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05aw; mkdir -p $O; cd $R; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --min-untimed 40 --shard-frames 16 --child > $O/pmc.log 2>&1)
python3 - $O/pmc <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print("columns:", sorted(rows[0].keys()) if rows else None)
per = collections.defaultdict(dict)
for r in rows:
    if "zj_fused" not in r["Kernel_Name"]: continue
    k = r["Dispatch_Id"]
    per[k][r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r: per[k]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ks = sorted(per, key=lambda x: int(x))[-30:]   # the warm ones
for name in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "ns"):
    v = [per[k][name] for k in ks if name in per[k]]
    if v: print(name, "mean of the last", len(v), "dispatches:", round(sum(v) / len(v), 1))
g = [per[k]["GRBM_GUI_ACTIVE"] / 8 / per[k]["ns"] for k in ks if "ns" in per[k] and "GRBM_GUI_ACTIVE" in per[k]]
if g: print("shader clock during the launch (GUI_ACTIVE / 8 XCDs / duration): mean %.3f GHz, min %.3f, max %.3f" % (sum(g) / len(g), min(g), max(g)))
v = [4 * per[k]["SQ_INSTS_VALU"] / 1024 / (per[k]["GRBM_GUI_ACTIVE"] / 8) for k in ks if "SQ_INSTS_VALU" in per[k] and "GRBM_GUI_ACTIVE" in per[k]]
if v: print("VALU issue cycles / active cycles: mean %.4f" % (sum(v) / len(v)))
PY
find $O -name "*.csv" -size +2M -delete
