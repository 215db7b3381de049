#!/bin/bash
# round 6, session b: the walker on the GPU box (v2, then the block-at-a-time decoder), the feeder A/B again with v2, and the
# reference files + device entropy tests against the regenerated goldens
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 300 python tools/walker_bench.py --pinned > $O/walker_v2.txt 2>&1
ZJ_WALKER_V1=1 timeout 300 python tools/walker_bench.py --pinned --no-pillow > $O/walker_v1.txt 2>&1
timeout 600 python tools/feeder_ab.py --quick > $O/feeder_ab_v2.txt 2>&1; echo "feeder rc $?"
timeout 600 python -m pytest tests/test_ref_images.py tests/test_gpu_entropy.py -q -x -m gpu > $O/gpu_ref_entropy.txt 2>&1; echo "pytest rc $?"
tail -3 $O/gpu_ref_entropy.txt; cat $O/walker_v2.txt $O/walker_v1.txt
