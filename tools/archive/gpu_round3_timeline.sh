#!/bin/bash
# Round 3: are ready blends held back by the hardware, or are they not ready earlier?  (rocprofv3 kernel trace, production pipeline)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=${1:-r03zx}
cd /tmp
for cfg in "chain:" "split_multi:--pipeline-mode split_multi --blend-streams 2"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  rm -rf /tmp/prof_$name
  ( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 5 --warmup 2 $extra > $O/${T}_bench_$name.json ) 2> /dev/null
  db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/timeline.py $db 25 > $O/${T}_timeline_$name.txt 2>&1
  echo "== $name"; head -4 $O/${T}_timeline_$name.txt; tail -5 $O/${T}_timeline_$name.txt
done
