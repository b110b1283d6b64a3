#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B --depth-sort radix > $O/r02n_radix.json ) 2> /dev/null
( timeout 200 $B --depth-sort bucket > $O/r02n_bucket.json ) 2> $O/r02n_bucket.err
( timeout 200 $B --depth-sort radix > $O/r02n_radix2.json ) 2> /dev/null
( timeout 200 $B --depth-sort bucket > $O/r02n_bucket2.json ) 2> /dev/null
( timeout 300 python -m pytest tests/test_gpu_render.py tests/test_gpu_graph_pipeline.py tests/test_gpu_parity_scale.py -m gpu -q -x 2>&1 | tail -4 ) > $O/r02n_pytest.log 2>&1
cd /tmp; rm -rf /tmp/prof_b
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 3 --warmup 2 --depth-sort bucket > /dev/null ) 2> /dev/null
db=$(find /tmp/prof_b -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/r02n_bucket_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for f in $O/r02n_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
cat $O/r02n_pytest.log; grep -E "k_bk_|k_radix|k_scan|k_preproc" $O/r02n_bucket_kernel_stats.csv
