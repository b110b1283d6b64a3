#!/bin/bash
# GPU box: configs[2] job time against cameras per launch sequence and batches in flight.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 6 --warmup 2"
for cfg in "1 4" "2 2" "2 3" "4 1" "4 2" "4 3" "8 1" "8 2" "3 2"; do
  set -- $cfg
  out=$(timeout 200 $B --camera-batch $1 --streams $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3f ms/step  blend alone %.3f ms  first_job %.1f ms' % (d['ms_per_step'], r['avg_launch_ms'], d['first_job_ms']))")
  echo "batch $1 streams $2: $out"
done
