#!/bin/bash
# round 5, session g: the evidence run -- bench tests, the full default line, the driver's command shapes (N = 1 and the
# 2-rank same-GPU dry run), the virtual-rank sweep over all 1024 frames of configs[4], rocprofv3 stats of the same command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q > $O/pytest_bench.log 2>&1; echo "bench tests exit $?" | tee -a $O/summary.txt
tail -6 $O/pytest_bench.log | tee -a $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_n1.json 2> $O/bench_driver_n1.err ) 2>&1 | grep real | tee -a $O/summary.txt
( time ZJ_BENCH_SAME_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err ) 2>&1 | grep real | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
for f in ("bench.json", "bench_driver_n1.json", "bench_2rank_same_gpu.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/r05g/" + f) if l.startswith("{")][-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "kernel", r["kernel_ms"], r["frac"], "per_rank_kernel_ms", r["per_rank_kernel_ms"], "golden", d["checksums_match_golden"],
              "frames", d["frames_checksummed"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "scattered", (r.get("scattered_batch") or {}).get("vs_kernel_ms"),
              (r.get("scattered_batch") or {}).get("adjacent_frames_vs_kernel_ms"), "one frame", (r.get("single_frame_launch") or {}).get("kernel_ms"), "dropped", r.get("live_counters_dropped"))
    except Exception as e:
        print(f, "FAILED", e)
PY
python tools/virtual_ranks.py 2>&1 | grep -v amdgpu.ids | tee $O/virtual_ranks.txt | tail -10 | tee -a $O/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-single-frame --no-live-traffic --no-e2e --no-dense-control --no-other-workloads > $O/prof_stats.log 2>&1)
grep -h '"metric"' $O/prof_stats.log | tail -1 | cut -c1-300 | tee -a $O/summary.txt
find $O/prof_stats -name "*kernel_stats*.csv" | head -1 | xargs -r head -6 | cut -c1-200 | tee -a $O/summary.txt
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -size +3M -delete
