#!/bin/bash
# Round-3 reference measurements on the final code: full GPU test suite, smoke, every bench workload, a 2-rank run on one GPU
# (gloo; validates the N > 1 code path of bench.py), stage times, rocprofv3 summaries (production streams and 1 stream).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03z}
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/${T}_pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1
( timeout 500 python bench.py > $O/${T}_bench_default.json ) 2> $O/${T}_bench_default.err
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B --workload sample --steps 20 > $O/${T}_bench_sample.json ) 2> /dev/null
( timeout 300 $B --workload render_cuda > $O/${T}_bench_render_cuda.json ) 2> /dev/null
( timeout 300 $B --t-floor 0 > $O/${T}_bench_exact.json ) 2> /dev/null
( timeout 400 python bench.py --no-cpu-baseline --workload config4 --steps 2 --warmup 1 > $O/${T}_bench_config4.json ) 2> $O/${T}_bench_config4.err
( G2PC_SHARE_GPU=1 G2PC_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --steps 3 --warmup 1 > $O/${T}_bench_w2_one_gpu_gloo.json ) 2> $O/${T}_bench_w2.err
( timeout 200 python tools/stage_times.py --every-job --jobs 4 > $O/${T}_stage_times.txt ) 2> /dev/null
( timeout 200 python tools/stage_times.py --workload sample > $O/${T}_stage_times_sample.txt ) 2> /dev/null
cd /tmp
for tag in prod:0 s1:1; do
  name=${tag%%:*}; st=${tag##*:}
  rm -rf /tmp/prof_$name
  extra=""; [ "$st" != "0" ] && extra="--streams $st"
  ( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 $extra > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof_$name.json ) 2> /dev/null
  db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/${T}_render_${name}_kernel_stats.csv
  [ -n "$db" ] && [ "$name" = "prod" ] && python $GRAFT_REPO_ROOT/tools/timeline.py $db 25 > $GRAFT_REPO_ROOT/$O/${T}_timeline_prod.txt 2>&1
done
rm -rf /tmp/prof_sample
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sample -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 --workload sample > /dev/null ) 2> /dev/null
db=$(find /tmp/prof_sample -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/${T}_sample_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cat $O/${T}_pytest.log $O/${T}_smoke.log
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-46s %.3e pts/s %.3f ms' % ('$f', d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
