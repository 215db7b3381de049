#!/bin/bash
# round 6, session q: the reference's benchmark images -- GPU parity of the two committed ones, the bench line with
# reference_bench, the walker on them by thread count
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ref_images.py tests/test_gpu_bench.py -q -x -m gpu > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -4 $O/tests.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r06q/bench.json') if l.startswith('{')][-1])
print(json.dumps(b.get('reference_bench'), indent=1))
PY
for t in 1 4 8 16; do timeout 300 python tools/walker_bench.py --pinned --no-pillow --threads $t --reps 7 2>&1 | grep -E "^speed_bench" | sed "s/^/threads $t: /"; done | tee $O/walker_speed_bench.txt
