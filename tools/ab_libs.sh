# ab_libs.sh lib1 lib2 ...: bench.py kernel time for each A/B library (GPU box), two interleaved passes
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  if [ -f zune-jpeg_amd/$lib ]; then
    ZJ_LIB=$lib python bench.py --no-cpu-baseline --no-live-traffic --no-e2e --shard-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=r['single_frame_launch']; print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'golden', d['checksums_match_golden'], '| one frame', s['kernel_ms'], s['frac'], '4 streams', s['four_streams_ms_per_frame'], '| dense', (r.get('dense_control') or {}).get('kernel_ms'))"
  fi
done; done
