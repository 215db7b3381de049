#!/bin/bash
# round 5, session af: zj_frame_desc.out_pitch (ABI 7) -- parity of the padded layout through every device entry point, the
# whole suite once more (the struct grew), then what a 128-byte-multiple pitch buys odd-pitch and ragged frames
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05af; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_pitch.py -m gpu -q -x > $O/pytest_pitch.log 2>&1; echo "pitch tests exit $?" | tee -a $O/summary.txt
tail -3 $O/pytest_pitch.log | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_pitch.py > $O/pytest.log 2>&1; echo "suite exit $?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/summary.txt
for rep in 1 2; do for pitch in 0 128; do
  echo "== ZJ_RAGGED_PITCH=$pitch" | tee -a $O/summary.txt
  ZJ_RAGGED_PITCH=$pitch ZJ_RAGGED_B=60 python tools/ragged_bench.py 2560x1792 2512x1792 2500x1786 1366x768 720x480 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
  ZJ_RAGGED_PITCH=$pitch python tools/ragged_bench.py 4080x4096 4090x4096 2>&1 | grep -v amdgpu.ids | grep "\->RGB" | cut -c1-130 | tee -a $O/summary.txt
done; done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?" | tee -a $O/summary.txt
python - $O/bench.json <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"], "golden", d.get("checksums_match_golden"))
for k, v in d["other_workloads"].items(): print(k, {a: v.get(a) for a in ("kernel_ms", "frac", "matches_wide_variant", "out_pitch", "rows_match_tight_layout", "error") if a in v})
PY
