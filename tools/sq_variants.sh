cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/sqv; mkdir -p $O
for v in onepass steal; do
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/$v -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --variant $v > $O/$v.log 2>&1)
echo "== $v"; python tools/pmc_summary.py $O/$v --tag $v 2>&1 | grep -E '"(SQ)|mean' | paste - -
done
