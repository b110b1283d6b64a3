#!/bin/bash
# Round 3: tile sort of the ~4 M instances as ONE 10-bit radix pass (--sort-bits 11) against two 5-bit passes (default), alternating.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zm}
for i in 1 2 3; do
  for v in 8 11; do
    ( timeout 200 python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --sort-bits $v > $O/${T}_bench_bits${v}_$i.json ) 2> /dev/null
  done
done
for f in $O/${T}_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3e pts/s %.3f ms' % ('$f'.split('/')[-1], d['value'], d['ms_per_step']))
except Exception as e: print('$f', str(e)[:80])
"; done
