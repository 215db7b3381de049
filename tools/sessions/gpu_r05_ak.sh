#!/bin/bash
# round 5, session ak: closing evidence on the final build (ABI 7: out_pitch; one-launch zeroing of the rows the strips never
# reach; the block waves' no-halo assumption): whole GPU suite + smoke, bench (default and the driver's shapes), kernel trace,
# counters, workloads, soaks with padded pitches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05ak; mkdir -p $O; cd $R; export TMPDIR=/tmp
bash tools/gpu_round.sh r05ak test smoke bench prof pmc sq
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
( time ZJ_BENCH_SAME_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err ) 2>&1 | grep real | tee -a $O/summary.txt
for f in bench bench_driver_n1 bench_2rank_same_gpu; do python - $O/$f.json <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["n_gpus"], d["value"], d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"),
      "golden", d.get("checksums_match_golden"), "cpu", d.get("cpu_baseline", {}).get("value"),
      "other", {k: v.get("frac") for k, v in (d.get("other_workloads") or {}).items()} if isinstance(d.get("other_workloads"), dict) else None)
PY
done
bash tools/workloads.sh $O/workloads.txt > /dev/null 2>&1; cat $O/workloads.txt | cut -c1-150 | tee -a $O/summary.txt
timeout 500 python tools/pixel_soak.py --seconds 400 --seed 77 > $O/pixel_soak.txt 2>&1; echo "pixel soak exit $?" | tee -a $O/summary.txt; head -24 $O/pixel_soak.txt | tee -a $O/summary.txt
timeout 300 python tools/pool_soak.py --seconds 150 --seed 9 --devices 0,0 > $O/pool_soak.txt 2>&1; echo "pool soak exit $?" | tee -a $O/summary.txt; tail -3 $O/pool_soak.txt | tee -a $O/summary.txt
