#!/bin/bash
# round 5, session e: A/B with the scatter flag folded into the base pointer and the table interleaved per frame;
# ragged widths: this build (fast interior + generic edge) against the round-4 path (generic everywhere), and a kernel
# trace that separates the two launches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R; export TMPDIR=/tmp
: > $O/summary.txt
for rep in 1 2 3; do for lib in libzjhip.so libzjhip_noscat.so; do
  ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --no-other-workloads --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['kernel_ms_single_launch'], s['frac'], '4 streams', s['four_streams_ms_per_frame'], '| dense', (r.get('dense_control') or {}).get('kernel_ms'))" | tee -a $O/summary.txt
done; done
python bench.py --no-cpu-baseline --no-e2e --no-live-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('full', d['value'], r['kernel_ms'], r['frac'], 'scattered', r['scattered_batch'])" | tee -a $O/summary.txt
echo "== ragged, this build" | tee -a $O/summary.txt
python tools/ragged_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
echo "== ragged, round-4 path (generic stores in every tile)" | tee -a $O/summary.txt
ZJ_LIB=libzjhip_noscat.so python tools/ragged_bench.py 2500x1786 4090x4096 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
echo "== kernel trace of the ragged sizes, this build" | tee -a $O/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged -o stats -- python3 $R/tools/ragged_bench.py 2500x1786 4090x4096 > $O/prof_ragged.log 2>&1)
find $O/prof_ragged -name "*kernel_stats*.csv" | head -1 | xargs -r head -12 | cut -c1-220 | tee -a $O/summary.txt
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -size +3M -delete
