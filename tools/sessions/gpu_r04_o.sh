#!/bin/bash
# round 4, session o: phase ablation of the final kernel (diagnostic build), timing of the default bench command
O=gpurun_out/r04o; mkdir -p $O
ZJ_LIB=libzjhip_ablate.so timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4 | tee $O/bench_time.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err ) 2>&1 | tail -4 | tee -a $O/bench_time.txt
python - <<'PY'
import json
for f in ("gpurun_out/r04o/bench.json", "gpurun_out/r04o/bench20.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d["roofline"]
    print(f, d["value"], d["steps"], d["ms_per_step"], r["kernel_ms"], r["kernel_launches_timed"], r["frac"], r["traffic"], r["traffic_replayed"])
PY
