#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B --workload sample --steps 20 > $O/r02f_sample20.json ) 2> /dev/null
( timeout 200 $B > $O/r02f_render.json ) 2> /dev/null
( timeout 300 python -m pytest tests/test_gpu_core.py tests/test_gpu_parity_scale.py -m gpu -q 2>&1 | tail -5 ) > $O/r02f_pytest.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 2 --warmup 1"
cd /tmp; rm -rf /tmp/pmcF /tmp/pmcW /tmp/prof_sample
( timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -o x -- $CMD > /dev/null ) 2> /dev/null
( timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -o x -- $CMD > /dev/null ) 2> /dev/null
f=$(find /tmp/pmcF -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmcW -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && [ -n "$w" ] && python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $f $w > $GRAFT_REPO_ROOT/$O/r02f_pmc_traffic.json
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sample -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2 --workload sample > /dev/null ) 2> /dev/null
db=$(find /tmp/prof_sample -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/r02f_sample_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for f in $O/r02f_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']), {k:round(v,3) for k,v in d['regions_ms_per_step'].items()})
except Exception as e: print('$f', str(e)[:60])
"; done
cat $O/r02f_pytest.log; head -6 $O/r02f_sample_kernel_stats.csv
