#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
for cap in 64 256 512; do
  ( timeout 200 $B --debug-walk-cap $cap > $O/r02i_cap$cap.json ) 2> /dev/null
done
( timeout 200 $B --debug-walk-cap 64 --streams 8 > $O/r02i_cap64_s8.json ) 2> /dev/null
( timeout 200 $B --debug-walk-cap 64 --streams 2 > $O/r02i_cap64_s2.json ) 2> /dev/null
( timeout 200 $B --debug-walk-cap 64 --streams 1 > $O/r02i_cap64_s1.json ) 2> /dev/null
( G2PC_BENCH_DEBUG=1 timeout 200 $B --debug-walk-cap 64 > /dev/null ) 2> $O/r02i_cap64_debug.txt
for f in $O/r02i_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']), {k:round(v,3) for k,v in d['regions_ms_per_step'].items() if 'raster' in k})
except Exception as e: print('$f', str(e)[:60])
"; done
cat $O/r02i_cap64_debug.txt | tail -6
