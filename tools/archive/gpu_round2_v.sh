#!/bin/bash
# session V: wave-per-bucket depth sort (8 KB LDS): bucket GPU test, A/B radix|bucket, overlap experiment, kernel trace of one chain
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 200 python -m pytest tests/test_gpu_core.py -k bucket -q 2>&1 | tail -2
for m in bucket radix bucket radix; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-parity --no-extra --no-cpu-baseline --depth-sort $m 2> $O/r02v.err | tee $O/r02v_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])" || tail -3 $O/r02v.err
done
timeout 200 python tools/experiments/blend_overlap.py 2>&1 | tail -6
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_v
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O/r02v_bench_under_rocprof.json ) 2> $O/r02v_rocprof.err
db=$(find /tmp/prof_v -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/r02v_render_s4_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/camera_chain.py $db 10 > $O/r02v_camera_chain.txt 2>&1
grep -E "k_bk_|k_blend|k_preprocess" $O/r02v_render_s4_kernel_stats.csv
