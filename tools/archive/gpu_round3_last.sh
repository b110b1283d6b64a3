#!/bin/bash
# Round 3, last commit: rocprofv3 summaries of every workload (production pipeline, sampling job, native-rasteriser job),
# the default bench line with parity / cpu_baseline, smoke.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zt}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/${T}_smoke.log 2>&1
( timeout 500 python bench.py > $O/${T}_bench_default.json ) 2> $O/${T}_bench_default.err
cd /tmp
for cfg in "prod:" "sample:--workload sample" "render_cuda:--workload render_cuda"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  rm -rf /tmp/prof_$name
  ( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 $extra > $O/${T}_bench_under_rocprof_$name.json ) 2> /dev/null
  db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/${T}_${name}_kernel_stats.csv
  [ -n "$db" ] && [ "$name" = "prod" ] && python $GRAFT_REPO_ROOT/tools/timeline.py $db 25 > $O/${T}_timeline_prod.txt 2>&1
done
cd $GRAFT_REPO_ROOT
cat $O/${T}_smoke.log
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-56s %.3e pts/s %.3f ms' % ('$f'.split('/')[-1], d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
head -5 $O/${T}_render_cuda_kernel_stats.csv | cut -c1-110
