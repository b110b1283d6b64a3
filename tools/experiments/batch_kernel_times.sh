#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) of the configs[2] job, one camera per launch sequence against four
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "1 1" "4 1" "4 2"; do
  set -- $cfg
  rm -rf /tmp/prof_b$1s$2
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$1s$2 -o x -- python $R/bench.py --no-parity --no-extra --no-cpu-baseline --steps 3 --warmup 2 --camera-batch $1 --streams $2 > $R/gpurun_out/r03l_bench_b$1s$2.json 2>/dev/null
  db=$(find /tmp/prof_b$1s$2 -name "*_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $R/gpurun_out/r03l_b$1s$2_kernel_stats.csv
done
