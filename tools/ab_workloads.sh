cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in 420-rgb 444-rgb 444-gray; do for lib in libzjhip.so libzjhip_oldk.so; do
ZJ_LIB=$lib python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $lib', d['roofline']['kernel_ms'])"
done; done; done
