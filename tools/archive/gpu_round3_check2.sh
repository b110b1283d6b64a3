#!/bin/bash
# Round 3: last check after the getter cache / fused filter gather: GPU tests that touch them, default + sample bench, stage times
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03zy}
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/${T}_pytest.log 2>&1
cat $O/${T}_pytest.log
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass"
for i in 1 2; do
( timeout 200 $B --steps 10 --warmup 3 > $O/${T}_bench_render_$i.json ) 2> /dev/null
( timeout 200 $B --workload sample --steps 20 --warmup 3 > $O/${T}_bench_sample_$i.json ) 2> /dev/null
done
( timeout 200 python tools/stage_times.py --every-job --jobs 3 > $O/${T}_stage_times.txt ) 2> /dev/null
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-46s %.3e pts/s %.3f ms' % ('$f', d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
tail -14 $O/${T}_stage_times.txt
