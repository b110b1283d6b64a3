#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( G2PC_POOL_SKIP_FIRST_JOBS=0 timeout 200 $B > $O/r02m_keep_first.json ) 2> /dev/null
for d in 1 2 3 4 5 8; do
  ( G2PC_POOL_SKIP_FIRST_JOBS=0 G2PC_DUMMY_STREAMS=$d timeout 200 $B > $O/r02m_keep_first_dummy$d.json ) 2> /dev/null
done
( G2PC_DUMMY_STREAMS=4 timeout 200 $B > $O/r02m_skip_first_dummy4.json ) 2> /dev/null
( timeout 200 $B > $O/r02m_base.json ) 2> /dev/null
for f in $O/r02m_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-46s %.3f ms first %.1f' % ('$f', d['ms_per_step'], d['first_job_ms']))
except Exception as e: print('$f', str(e)[:80])
"; done
