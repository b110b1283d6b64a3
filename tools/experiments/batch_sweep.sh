#!/bin/bash
# GPU box: configs[2] job time against pipeline mode, cameras per launch sequence and batches in flight.
# usage: batch_sweep.sh "mode batch slots;mode batch slots;..."
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 6 --warmup 2"
CFGS=${1:-"chain 1 4;chain 2 4;chain 4 4;chain 2 3;chain 4 2"}
IFS=';' read -ra LIST <<< "$CFGS"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  out=$(timeout 200 $B --pipeline-mode $1 --camera-batch $2 --streams $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  first_job %.1f ms' % (d['ms_per_step'], d['first_job_ms']))")
  echo "$1 batch $2 slots $3: $out"
done
