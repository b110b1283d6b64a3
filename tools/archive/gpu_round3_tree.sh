#!/bin/bash
# Round 3: data-dependent quad-tree (k_tile_gate, per-camera passes).  GPU tests (all), three runs of the default bench
# to see that the gate in place of k_check_tile_load leaves the headline where it was, kernel stats of the production job.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zr}
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_gpu_tests.log 2>&1
for i in 1 2 3; do
  ( timeout 300 python bench.py --no-parity --no-extra --no-cpu-baseline > $O/${T}_bench_$i.json ) 2> $O/${T}_bench_$i.err
done
cd /tmp
rm -rf /tmp/prof_prod
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_prod -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 > $O/${T}_bench_under_rocprof.json ) 2> /dev/null
db=$(find /tmp/prof_prod -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/${T}_prod_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cat $O/${T}_gpu_tests.log
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-44s %.3e pts/s %.3f ms' % ('$f'.split('/')[-1], d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
grep -i "gate\|blend_py_dl\|preprocess_py\|tile_ranges" $O/${T}_prod_kernel_stats.csv | cut -c1-160
