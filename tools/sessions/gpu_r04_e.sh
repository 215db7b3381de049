#!/bin/bash
# round 4, session e: whole GPU suite on the round's kernels, persistent-workgroup lab experiment, A/B
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
timeout 900 python tools/persist_lab.py 60 2>&1 | grep -v "amdgpu.ids" | tee $O/persist.txt
timeout 600 bash tools/ab_libs.sh libzjhip_base.so libzjhip.so libzjhip_base.so libzjhip.so 2>&1 | tee $O/ab.txt
