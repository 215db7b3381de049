# ab_libs_w.sh workload lib1 lib2 ...: bench.py --workload W kernel time for each A/B library (GPU box)
cd $GRAFT_REPO_ROOT
W=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  if [ -f zune-jpeg_amd/$lib ]; then
    ZJ_LIB=$lib python bench.py --no-cpu-baseline --workload $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac'])"
  fi
done; done
