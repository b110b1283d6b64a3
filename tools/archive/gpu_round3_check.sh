#!/bin/bash
# Round 3 mid-session check: full GPU test suite, smoke, default bench + sampler bench.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03za}
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/${T}_pytest.log 2>&1
cat $O/${T}_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/${T}_smoke.log 2>&1
cat $O/${T}_smoke.log
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass"
( timeout 200 $B --workload sample --steps 20 --warmup 3 > $O/${T}_bench_sample.json ) 2> $O/${T}_bench_sample.err
( timeout 200 $B --steps 10 --warmup 3 > $O/${T}_bench_render.json ) 2> $O/${T}_bench_render.err
( timeout 200 python tools/stage_times.py --workload sample > $O/${T}_stage_times_sample.txt ) 2> /dev/null
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-46s %.3e pts/s %.3f ms' % ('$f', d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
tail -8 $O/${T}_stage_times_sample.txt
