#!/bin/bash
# final reference of round 2 after the deferred colour resolve: full GPU suite, smoke, default line, rocprofv3 summary (4 streams)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; T=r02z3
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > $O/${T}_pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1
( timeout 500 python bench.py > $O/${T}_bench_default.json ) 2> $O/${T}_bench_default.err
cd /tmp; rm -rf /tmp/prof_s4
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s4 -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof_s4.json ) 2> /dev/null
db=$(find /tmp/prof_s4 -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/${T}_render_s4_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cat $O/${T}_pytest.log $O/${T}_smoke.log
python -c "
import json
d=json.load(open('$O/${T}_bench_default.json')); print('default %.3e pts/s %.3f ms' % (d['value'], d['ms_per_step'])); print({k: d['parity'][k] for k in ('mask_flips','contrib_frac_gt_1e-4','image_frac_gt_1e-4','ppg_equal','sample_points')})
"
head -8 $O/${T}_render_s4_kernel_stats.csv
