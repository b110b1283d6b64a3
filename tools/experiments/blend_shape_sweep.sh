#!/bin/bash
# GPU box: configs[2] job time against the blend's work decomposition (8x8 sub-blocks per wave, kernel variant)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 6 --warmup 2"
for cfg in "2 1 2 4" "1 1 2 4" "1 1 1 4" "4 1 2 4" "2 0 2 4" "2 1 1 4" "1 1 4 3"; do
  set -- $cfg
  out=$(timeout 200 $B --blend-subblocks $1 --blend-variant $2 --camera-batch $3 --streams $4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f ms/step  blend alone %.3f ms' % (d['ms_per_step'], r['avg_launch_ms'] if r else -1))")
  echo "subblocks $1 variant $2 batch $3 slots $4: $out"
done
