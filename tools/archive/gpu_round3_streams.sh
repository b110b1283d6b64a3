#!/bin/bash
# Round 3: do two blends in flight pay?  Split pipeline modes (heads on high-priority streams) with 1 .. 4 blend streams.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03zb}
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 10 --warmup 3"
run() {
  name=$1; shift
  ( timeout 200 $B "$@" > $O/${T}_bench_$name.json ) 2> $O/${T}_bench_$name.err
  python -c "
import json
try:
    d=json.load(open('$O/${T}_bench_$name.json')); print('%-34s %.3f ms/job  %.3e pts/s' % ('$name', d['ms_per_step'], d['value']))
except Exception as e: print('$name', 'FAILED', str(e)[:100])
"
}
run chain_b2_s4
run splitmulti_bs1_b2_s4   --pipeline-mode split_multi --blend-streams 1
run splitmulti_bs2_b2_s4   --pipeline-mode split_multi --blend-streams 2
run splitmulti_bs2_b2_s6   --pipeline-mode split_multi --blend-streams 2 --streams 6
run splitmulti_bs2_b2_s8   --pipeline-mode split_multi --blend-streams 2 --streams 8
run splitmulti_bs2_b1_s8   --pipeline-mode split_multi --blend-streams 2 --streams 8 --camera-batch 1
run splitmulti_bs3_b2_s6   --pipeline-mode split_multi --blend-streams 3 --streams 6
run splitmulti_bs4_b1_s8   --pipeline-mode split_multi --blend-streams 4 --streams 8 --camera-batch 1
run split_bs2_b2_s4        --pipeline-mode split --blend-streams 2
run splitmulti_bs2_b4_s4   --pipeline-mode split_multi --blend-streams 2 --camera-batch 4
run chain_b2_s4_again
