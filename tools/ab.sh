cd $GRAFT_REPO_ROOT
if [ "${1:-}" = "test" ]; then timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3; fi
for v in onepass compact persistent onepass; do python bench.py --no-cpu-baseline --variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"; done
