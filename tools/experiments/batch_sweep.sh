#!/bin/bash
# GPU box: configs[2] job time against pipeline mode, cameras per launch sequence and batches in flight.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 6 --warmup 2"
for cfg in ${1:-"chain 1 4" "split 1 4" "split 2 3" "split 4 2" "split 4 3" "split 8 2" "split 8 3" "split 6 3" "chain 2 3"}; do
  set -- $cfg
  out=$(timeout 200 $B --pipeline-mode $1 --camera-batch $2 --streams $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  first_job %.1f ms' % (d['ms_per_step'], d['first_job_ms']))")
  echo "$1 batch $2 slots $3: $out"
done
