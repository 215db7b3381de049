# workloads.sh [outfile]: one bench line summary per workload (GPU box): value, ms/step, kernel ms, GB/s, fraction
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/workloads.txt}
: > $OUT
for w in 420-rgb 444-rgb 444-gray 422-rgb 440-rgb 420-rgba 420-chw; do
  python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w', d['value'], 'MP/s', d['ms_per_step'], 'ms/step', r['kernel_ms'], 'ms/launch', r['achieved'], 'GB/s', r['frac'])" | tee -a $OUT
done
for w in 420-rgb-2500x1786 420-rgb-2500x1786-pitch128 444-rgb-2500x1786; do  # ragged width: 60 frames per launch = the pixels of 16 frames of 4096 x 4096
  python bench.py --no-cpu-baseline --workload $w --frames 60 --shard-frames 120 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w', d['value'], 'MP/s', d['ms_per_step'], 'ms/step', r['kernel_ms'], 'ms/launch', r['achieved'], 'GB/s', r['frac'], r['kernel'][10:50])" | tee -a $OUT
done
for v in packed wide packed-direct packed; do
  python bench.py --no-cpu-baseline --variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('420-rgb variant $v', d['value'], 'MP/s', r['kernel_ms'], 'ms/launch', r['achieved'], 'GB/s', r['frac'])" | tee -a $OUT
done
