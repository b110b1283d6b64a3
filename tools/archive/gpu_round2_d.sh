#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B > $O/r02d_base.json ) 2> /dev/null
for q in 8 16; do for st in 4 6 8; do
  ( GPU_MAX_HW_QUEUES=$q timeout 200 $B --streams $st > $O/r02d_q${q}_s$st.json ) 2> /dev/null
done; done
( GPU_MAX_HW_QUEUES=8 timeout 200 $B --blend-subblocks 1 --streams 8 > $O/r02d_q8_s8_sub1.json ) 2> /dev/null
( timeout 200 $B --blend-subblocks 1 > $O/r02d_sub1.json ) 2> /dev/null
( timeout 200 $B --sort-bits 11 > $O/r02d_sort11.json ) 2> /dev/null
( timeout 200 $B --workload sample > $O/r02d_sample.json ) 2> /dev/null
( timeout 400 python -m pytest tests/test_gpu_parity_scale.py -m gpu -q 2>&1 | tail -5 ) > $O/r02d_pytest_parity.log 2>&1
for f in $O/r02d_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('%-40s %.2f ms' % ('$f', d['ms_per_step']))
except Exception as e: print('$f ERR')
"; done
cat $O/r02d_pytest_parity.log
