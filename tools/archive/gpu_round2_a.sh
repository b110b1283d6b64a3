#!/bin/bash
# Round-2 GPU session A: tests, default bench (with parity), blend A/B, streams sweep, sample workload, rocprof summaries.
# Everything lands in gpurun_out/r02a_*; each step has its own timeout so that a hang cannot eat the box.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
nproc > $O/r02a_nproc.txt
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity_scale.py 2>&1 | tail -15 ) > $O/r02a_pytest.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_parity_scale.py -m gpu -q -s 2>&1 | tail -30 ) > $O/r02a_pytest_parity.log 2>&1
( timeout 600 python bench.py > $O/r02a_bench_default.json ) 2> $O/r02a_bench_default.err
for v in 0 1 2; do
  ( timeout 200 $B --blend-variant $v > $O/r02a_ab_variant$v.json ) 2> $O/r02a_ab_variant$v.err
done
for st in 2 6 8; do
  ( timeout 200 $B --streams $st > $O/r02a_ab_streams$st.json ) 2> $O/r02a_ab_streams$st.err
done
( timeout 200 $B --blend-variant 0 --streams 8 > $O/r02a_ab_variant0_streams8.json ) 2> /dev/null
( timeout 200 $B --t-floor 0 > $O/r02a_bench_exact.json ) 2> $O/r02a_bench_exact.err
( timeout 200 $B --workload sample > $O/r02a_bench_sample.json ) 2> $O/r02a_bench_sample.err
( timeout 300 $B --workload render_cuda > $O/r02a_bench_render_cuda.json ) 2> $O/r02a_bench_render_cuda.err
( G2PC_POOL_SKIP_FIRST_JOBS=0 timeout 200 $B > $O/r02a_ab_pool_keep_first.json ) 2> /dev/null
# rocprof summaries: overlapped (default 4 streams) and --streams 1, same commit
cd /tmp
for tag in s4:4 s1:1; do
  name=${tag%%:*}; st=${tag##*:}
  rm -rf /tmp/prof_$name
  ( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2 --streams $st > $GRAFT_REPO_ROOT/$O/r02a_bench_under_rocprof_$name.json ) 2> $GRAFT_REPO_ROOT/$O/r02a_rocprof_$name.err
  db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/r02a_render_${name}_kernel_stats.csv 2>> $GRAFT_REPO_ROOT/$O/r02a_rocprof_$name.err
done
rm -rf /tmp/prof_sample
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sample -o x -- python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2 --workload sample > /dev/null ) 2> $GRAFT_REPO_ROOT/$O/r02a_rocprof_sample.err
db=$(find /tmp/prof_sample -name "*_results.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $GRAFT_REPO_ROOT/$O/r02a_sample_kernel_stats.csv
cd $GRAFT_REPO_ROOT
( timeout 200 python tools/chunk_work.py > $O/r02a_chunk_work.txt ) 2> $O/r02a_chunk_work.err
ls -la $O | tail -40
