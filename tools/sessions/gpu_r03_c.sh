#!/bin/bash
# Round 3, session C: what a ONE-frame launch costs -- the kernel's own duration (rocprofv3 begin..end) against the
# back-to-back rate HIP events see (which includes the gap between dependent dispatches of one stream)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R; export TMPDIR=/tmp
S=$O/summary.txt; : > $S
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o stats -- python3 $R/bench.py --frames 1 --shard-frames 16 --steps 3000 --warmup 500 --no-cpu-baseline --no-single-frame --no-live-traffic > $O/prof1.log 2>&1)
grep -h '"metric"' $O/prof1.log | tail -1 | cut -c1-300 | tee -a $S
find $O/prof1 -name "*kernel_stats*.csv" | head -1 | xargs -r head -3 | cut -c1-200 | tee -a $S
python3 - <<PY | tee -a $S
import csv,glob
f=glob.glob("$O/prof1/**/*kernel_trace*.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "zj_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[600:]
dur=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
gap=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(rows,rows[1:])]
per=[int(b["Start_Timestamp"])-int(a["Start_Timestamp"]) for a,b in zip(rows,rows[1:])]
import statistics as st
print("launches",len(rows),"duration ns: mean %.0f median %.0f min %d"%(st.mean(dur),st.median(dur),min(dur)))
print("gap ns: mean %.0f median %.0f"%(st.mean(gap),st.median(gap)),"start-to-start ns: mean %.0f median %.0f"%(st.mean(per),st.median(per)))
PY
find $O -name "*.csv" -size +3M -delete
