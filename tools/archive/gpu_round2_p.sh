#!/bin/bash
# session P: deterministic bucket depth sort (no global atomics): fault probe, primitive test, A/B of the headline job, kernel profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo N=1000000 async; timeout 90 python tools/experiments/bucket_fault_probe.py 1000000 50 4 2>&1 | tail -2
timeout 200 python -m pytest tests/test_gpu_core.py -k bucket -q 2>&1 | tail -2
for m in radix bucket radix bucket; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-parity --no-extra --no-cpu-baseline --depth-sort $m 2> gpurun_out/r02p_$m.err | tee gpurun_out/r02p_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])" || tail -3 gpurun_out/r02p_$m.err
done
cd /tmp; export TMPDIR=/tmp
for m in bucket; do
  rm -rf /tmp/prof_$m
  ( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-parity --no-extra --no-cpu-baseline --streams 1 --depth-sort $m > /dev/null ) 2> /dev/null
  db=$(find /tmp/prof_$m -name "*_results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/gpurun_out/r02p_render_s1_${m}_kernel_stats.csv
  head -24 $GRAFT_REPO_ROOT/gpurun_out/r02p_render_s1_${m}_kernel_stats.csv
done
