#!/bin/bash
# Round 3: the split pipeline modes shared hardware queues (6 streams on 4 queues: the two blend streams never overlapped).
# The same sweep with GPU_MAX_HW_QUEUES raised.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03zw}
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 10 --warmup 3"
run() {
  name=$1; shift
  ( env "$@" > $O/${T}_bench_$name.json ) 2> $O/${T}_bench_$name.err
  python -c "
import json
try:
    d=json.load(open('$O/${T}_bench_$name.json')); print('%-40s %.3f ms/job  %.3e pts/s' % ('$name', d['ms_per_step'], d['value']))
except Exception as e: print('$name', 'FAILED', str(e)[:100])
"
}
run chain_q4                 timeout 200 $B
run chain_q8                 GPU_MAX_HW_QUEUES=8 timeout 200 $B
run splitmulti_bs2_s4_q8     GPU_MAX_HW_QUEUES=8 timeout 200 $B --pipeline-mode split_multi --blend-streams 2
run splitmulti_bs2_s4_q12    GPU_MAX_HW_QUEUES=12 timeout 200 $B --pipeline-mode split_multi --blend-streams 2
run splitmulti_bs2_s6_q12    GPU_MAX_HW_QUEUES=12 timeout 200 $B --pipeline-mode split_multi --blend-streams 2 --streams 6
run splitmulti_bs3_s6_q12    GPU_MAX_HW_QUEUES=12 timeout 200 $B --pipeline-mode split_multi --blend-streams 3 --streams 6
run splitmulti_bs1_s4_q8     GPU_MAX_HW_QUEUES=8 timeout 200 $B --pipeline-mode split_multi --blend-streams 1
run splitmulti_bs4_s4_q8     GPU_MAX_HW_QUEUES=8 timeout 200 $B --pipeline-mode split_multi --blend-streams 4
run splitmulti_bs2_b1_s8_q16 GPU_MAX_HW_QUEUES=16 timeout 200 $B --pipeline-mode split_multi --blend-streams 2 --streams 8 --camera-batch 1
run split_bs2_s4_q8          GPU_MAX_HW_QUEUES=8 timeout 200 $B --pipeline-mode split --blend-streams 2
run chain_q4_again           timeout 200 $B
cd /tmp
rm -rf /tmp/prof_q
( GPU_MAX_HW_QUEUES=12 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 --pipeline-mode split_multi --blend-streams 2 > /dev/null ) 2> /dev/null
db=$(find /tmp/prof_q -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/timeline.py $db 25 > $GRAFT_REPO_ROOT/$O/${T}_timeline_splitmulti_bs2_q12.txt 2>&1
head -4 $GRAFT_REPO_ROOT/$O/${T}_timeline_splitmulti_bs2_q12.txt; tail -5 $GRAFT_REPO_ROOT/$O/${T}_timeline_splitmulti_bs2_q12.txt
