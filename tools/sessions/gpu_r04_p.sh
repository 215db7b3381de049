#!/bin/bash
# round 4, session p: bench with the clock pre-warm; rocprofv3 --stats of the same command (average vs HIP events)
O=gpurun_out/r04p; mkdir -p $O
bash tools/gpu_round.sh r04p/round bench prof > $O/round.log 2>&1
python - <<'PY'
import json, csv, glob
d = json.loads([l for l in open("gpurun_out/r04p/round/bench.json") if l.startswith("{")][-1]); r = d["roofline"]
print("bench:", d["value"], d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "dense", r["dense_control"]["kernel_ms"], "one frame", r["single_frame_launch"]["kernel_ms"])
for f in glob.glob("gpurun_out/r04p/round/prof_stats/*kernel_stats*.csv"):
    for row in csv.DictReader(open(f)):
        if "zj_fused" in row["Name"]:
            print("rocprofv3 stats:", row["Calls"], "calls avg", float(row["AverageNs"]) / 1e3, "us min", float(row["MinNs"]) / 1e3, "max", float(row["MaxNs"]) / 1e3)
l = [x for x in open("gpurun_out/r04p/round/prof_stats.log") if x.startswith("{")]
if l:
    d = json.loads(l[-1]); print("same command, HIP events:", d["roofline"]["kernel_ms"])
PY
