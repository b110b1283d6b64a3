#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B > $O/r02l_base.json ) 2> /dev/null
( timeout 200 $B --sort-single-pass-bits 10 > $O/r02l_tile1pass.json ) 2> /dev/null
( timeout 200 $B --sort-single-pass-bits 10 --streams 1 > $O/r02l_tile1pass_s1.json ) 2> /dev/null
( timeout 200 $B --sort-single-pass-bits 10 --debug-walk-cap 64 > $O/r02l_tile1pass_cap64.json ) 2> /dev/null
for f in $O/r02l_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']), {k:round(v,3) for k,v in d['regions_ms_per_step'].items() if 'raster' in k})
except Exception as e: print('$f', str(e)[:60])
"; done
