#!/bin/bash
# round 5, session ar: the final build once more -- whole GPU suite + smoke, and the bench lines with roofline.same_box_copy
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05ar; mkdir -p $O; cd $R; export TMPDIR=/tmp
bash tools/gpu_round.sh r05ar test smoke bench
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
( time ZJ_BENCH_SAME_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err ) 2>&1 | grep real | tee -a $O/summary.txt
for f in bench bench_driver_n1 bench_2rank_same_gpu; do python - $O/$f.json <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["n_gpus"], d["value"], d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"),
      "golden", d.get("checksums_match_golden"), "cpu", d.get("cpu_baseline", {}).get("value"), "copy", r.get("same_box_copy"),
      "other", {k: v.get("frac") for k, v in (d.get("other_workloads") or {}).items()} if isinstance(d.get("other_workloads"), dict) else None)
PY
done
