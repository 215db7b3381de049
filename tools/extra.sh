cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-extra}; mkdir -p $O
for w in 420-rgb 444-rgb 444-gray; do python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tee -a $O/workloads.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], 'MP/s', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'GB/s', d['roofline']['frac'])"; done
python tools/e2e.py 2>/dev/null | tee $O/e2e.txt
