#!/bin/bash
# GPU box: kernel-trace timeline analysis (tools/timeline.py, tools/stream_gaps.py) of the configs[2] job for a pipeline configuration
# usage: timeline_run.sh "<mode> <batch> <slots>" tag
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
set -- $1 $2
rm -rf /tmp/prof_tl
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o x -- python $R/bench.py --no-parity --no-extra --no-cpu-baseline --steps 3 --warmup 2 --no-profile-pass --pipeline-mode $1 --camera-batch $2 --streams $3 > /dev/null 2>&1
db=$(find /tmp/prof_tl -name "*_results.db" | head -1)
echo "== $1 batch $2 slots $3"
python $R/tools/timeline.py $db $((50 / $2)) 
python $R/tools/stream_gaps.py $db $((50 / $2))
