#!/bin/bash
# Round-2 GPU session H: tail diagnostics (walk cap), then the round's reference measurements on the final code.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B > $O/r02h_base.json ) 2> /dev/null
for cap in 1024 1536 2048 4096; do
  ( timeout 200 $B --debug-walk-cap $cap > $O/r02h_cap$cap.json ) 2> /dev/null
  ( timeout 200 $B --debug-walk-cap $cap --streams 1 > $O/r02h_cap${cap}_s1.json ) 2> /dev/null
done
( timeout 200 $B --streams 1 > $O/r02h_base_s1.json ) 2> /dev/null
for f in $O/r02h_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']), {k:round(v,3) for k,v in d['regions_ms_per_step'].items() if 'raster' in k})
except Exception as e: print('$f', str(e)[:60])
"; done
