#!/bin/bash
# GPU box: rocprofv3 per-kernel stats of the working tree (batch 1, 4 slots, chain) against a second tree under _ab_old/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
A="--no-parity --no-extra --no-cpu-baseline --steps 4 --warmup 2"
rm -rf /tmp/prof_old /tmp/prof_new
( cd $R/_ab_old && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_old -o x -- python bench.py $A > $R/gpurun_out/r03o_old.json 2>/dev/null )
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_new -o x -- python bench.py $A --pipeline-mode chain --camera-batch 1 --streams 4 > $R/gpurun_out/r03o_new.json 2>/dev/null )
for t in old new; do
  db=$(find /tmp/prof_$t -name "*_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/gpurun_out/r03o_${t}_kernel_stats.csv
  python -c "import json; d=json.load(open('$R/gpurun_out/r03o_$t.json')); print('$t', d['ms_per_step'])"
  head -22 $R/gpurun_out/r03o_${t}_kernel_stats.csv
done
