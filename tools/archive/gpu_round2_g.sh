#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for fl in 1e-6; do
  ( timeout 200 $B --t-floor $fl > $O/r02g_floor_$fl.json ) 2> /dev/null
done
( timeout 200 $B --workload sample > $O/r02g_sample.json ) 2> /dev/null
( timeout 200 $B --workload sample --steps 20 > $O/r02g_sample20.json ) 2> /dev/null
( timeout 400 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_core.py -m gpu -q 2>&1 | tail -5 ) > $O/r02g_pytest.log 2>&1
cd /tmp; rm -rf /tmp/prof_sample
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sample -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2 --workload sample > /dev/null ) 2> /dev/null
db=$(find /tmp/prof_sample -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/r02g_sample_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for f in $O/r02g_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']), {k:round(v,3) for k,v in d['regions_ms_per_step'].items()})
except Exception as e: print('$f ERR')
"; done
cat $O/r02g_pytest.log; head -8 $O/r02g_sample_kernel_stats.csv
