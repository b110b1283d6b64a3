#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02j}
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 200 $B > $O/${TAG}_base.json ) 2> /dev/null
( timeout 200 $B --debug-walk-cap 64 > $O/${TAG}_cap64.json ) 2> /dev/null
( timeout 200 $B --streams 1 > $O/${TAG}_s1.json ) 2> /dev/null
( timeout 200 $B --workload sample > $O/${TAG}_sample.json ) 2> /dev/null
( timeout 300 python -m pytest tests/test_gpu_core.py tests/test_gpu_render.py tests/test_gpu_parity_scale.py -m gpu -q -x 2>&1 | tail -4 ) > $O/${TAG}_pytest.log 2>&1
for f in $O/${TAG}_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('%-40s %.3f ms' % ('$f', d['ms_per_step']), {k:round(v,3) for k,v in d['regions_ms_per_step'].items() if 'raster' in k})
except Exception as e: print('$f', str(e)[:60])
"; done
cat $O/${TAG}_pytest.log
