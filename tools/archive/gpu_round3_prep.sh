#!/bin/bash
# Round 3: preprocess with binary-searched interval ranges: parity tests, rocprofv3 of one stream (kernel alone), job
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zu}
( timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_parity_scale.py tests/test_gpu_graph_pipeline.py tests/test_gpu_helpers.py -m gpu -q 2>&1 | tail -4 ) > $O/${T}_pytest.log 2>&1
cat $O/${T}_pytest.log
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass"
for i in 1 2 3; do
( timeout 200 $B --steps 10 --warmup 3 > $O/${T}_bench_render_$i.json ) 2> /dev/null
python -c "
import json
d=json.load(open('$O/${T}_bench_render_$i.json')); print('run $i  %.3f ms/job  %.3e pts/s' % (d['ms_per_step'], d['value']))
"
done
cd /tmp
rm -rf /tmp/prof_s1
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s1 -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 --streams 1 > /dev/null ) 2> /dev/null
db=$(find /tmp/prof_s1 -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/${T}_render_s1_kernel_stats.csv
head -6 $O/${T}_render_s1_kernel_stats.csv | cut -c1-120
