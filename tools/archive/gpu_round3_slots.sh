#!/bin/bash
# Round 3: batch slots per stream (the stream's next batch queued while the previous one runs: no host gap between them)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03zv}
( timeout 400 python -m pytest tests/test_gpu_graph_pipeline.py tests/test_gpu_parity_scale.py -m gpu -q -x 2>&1 | tail -4 ) > $O/${T}_pytest.log 2>&1
cat $O/${T}_pytest.log
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 10 --warmup 3"
run() {
  name=$1; shift
  ( timeout 200 $B "$@" > $O/${T}_bench_$name.json ) 2> $O/${T}_bench_$name.err
  python -c "
import json
try:
    d=json.load(open('$O/${T}_bench_$name.json')); print('%-28s %.3f ms/job  %.3e pts/s' % ('$name', d['ms_per_step'], d['value']))
except Exception as e: print('$name', 'FAILED', str(e)[:100])
"
}
run s4_p1   --slots-per-stream 1
run s4_p2   --slots-per-stream 2
run s4_p3   --slots-per-stream 3
run s3_p2   --slots-per-stream 2 --streams 3
run s2_p2   --slots-per-stream 2 --streams 2
run s6_p2   --slots-per-stream 2 --streams 6
run s4_p2_b1 --slots-per-stream 2 --camera-batch 1
run s4_p2_b3 --slots-per-stream 2 --camera-batch 3
run s4_p2_b4 --slots-per-stream 2 --camera-batch 4
run s4_p1_again --slots-per-stream 1
run s4_p2_again --slots-per-stream 2
