#!/bin/bash
# Round 3: cost of k_tile_gate in the production pipeline (two bench runs + rocprofv3 kernel stats), quad-tree GPU tests.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zo}
( timeout 600 python -m pytest tests/test_gpu_quadtree.py tests/test_gpu_graph_pipeline.py -q 2>&1 | tail -4 ) > $O/${T}_quadtree_tests.log 2>&1
for i in 1 2; do
  ( timeout 300 python bench.py --no-parity --no-extra --no-cpu-baseline > $O/${T}_bench_$i.json ) 2> $O/${T}_bench_$i.err
done
cd /tmp
rm -rf /tmp/prof_prod
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_prod -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 > $O/${T}_bench_under_rocprof.json ) 2> /dev/null
db=$(find /tmp/prof_prod -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/${T}_prod_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cat $O/${T}_quadtree_tests.log
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-44s %.3e pts/s %.3f ms' % ('$f'.split('/')[-1], d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
grep -i "gate\|tile_ranges\|resolve_count" $O/${T}_prod_kernel_stats.csv | cut -c1-160
