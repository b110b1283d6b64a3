#!/bin/bash
# session T: where does a pipelined camera spend its time now (cull + dual list + bucket sort)?  kernel trace -> timeline, gaps, one chain
cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
rm -rf /tmp/prof_t
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O/r02t_bench_under_rocprof.json ) 2> $O/r02t_rocprof.err
db=$(find /tmp/prof_t -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $O/r02t_render_s4_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/timeline.py $db 50 > $O/r02t_timeline.txt 2>&1
python $GRAFT_REPO_ROOT/tools/stream_gaps.py $db 50 > $O/r02t_stream_gaps.txt 2>&1
python $GRAFT_REPO_ROOT/tools/camera_chain.py $db 10 > $O/r02t_camera_chain.txt 2>&1
python $GRAFT_REPO_ROOT/tools/camera_chain.py $db 23 >> $O/r02t_camera_chain.txt 2>&1
head -30 $O/r02t_render_s4_kernel_stats.csv
