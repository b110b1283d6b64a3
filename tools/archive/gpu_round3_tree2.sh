#!/bin/bash
# Round 3: the quad-tree GPU tests alone (verbose results), then the whole GPU suite and two bench runs.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zp}
( timeout 600 python -m pytest tests/test_gpu_quadtree.py -q -s 2>&1 | grep -v "^$" | tail -40 ) > $O/${T}_quadtree_tests.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/${T}_gpu_tests.log 2>&1
for i in 1 2; do
  ( timeout 300 python bench.py --no-parity --no-extra --no-cpu-baseline > $O/${T}_bench_$i.json ) 2> $O/${T}_bench_$i.err
done
cd /tmp
rm -rf /tmp/prof_prod
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_prod -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 > $O/${T}_bench_under_rocprof.json ) 2> /dev/null
db=$(find /tmp/prof_prod -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/${T}_prod_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cut -c1-400 $O/${T}_quadtree_tests.log
cat $O/${T}_gpu_tests.log
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-44s %.3e pts/s %.3f ms' % ('$f'.split('/')[-1], d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
grep -i "gate\|tile_ranges" $O/${T}_prod_kernel_stats.csv | cut -c1-160
