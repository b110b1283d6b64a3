#!/bin/bash
# Round-2 GPU session B: parity diagnostics, front-priority A/B, allocator experiment, sample, stage times, new tests.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 5 --warmup 2"
( timeout 300 python tools/experiments/parity_diag.py > $O/r02b_parity_diag.json ) 2> $O/r02b_parity_diag.err
( timeout 200 $B > $O/r02b_ab_prio_v1.json ) 2> $O/r02b_ab_prio_v1.err
( timeout 200 $B --blend-variant 0 > $O/r02b_ab_prio_v0.json ) 2> /dev/null
( timeout 200 $B --blend-variant 0 --no-front-priority > $O/r02b_ab_noprio_v0.json ) 2> /dev/null
( timeout 200 $B --blend-variant 0 --streams 6 > $O/r02b_ab_prio_v0_s6.json ) 2> /dev/null
( timeout 200 $B --blend-variant 0 --streams 3 > $O/r02b_ab_prio_v0_s3.json ) 2> /dev/null
( G2PC_POOL_SKIP_FIRST_JOBS=0 G2PC_PREALLOC_GB=6 timeout 200 $B --blend-variant 0 > $O/r02b_ab_prealloc_keep_first.json ) 2> /dev/null
( G2PC_POOL_SKIP_FIRST_JOBS=0 timeout 200 $B --blend-variant 0 > $O/r02b_ab_keep_first.json ) 2> /dev/null
( timeout 200 $B --workload sample > $O/r02b_bench_sample.json ) 2> $O/r02b_bench_sample.err
( timeout 300 python -m pytest tests/test_gpu_config4.py tests/test_gpu_helpers.py tests/test_gpu_cuda_semantics.py -m gpu -q 2>&1 | tail -15 ) > $O/r02b_pytest_new.log 2>&1
( timeout 200 python tools/stage_times.py > $O/r02b_stage_times.txt ) 2> $O/r02b_stage_times.err
( timeout 200 python tools/stage_times.py --workload sample > $O/r02b_stage_times_sample.txt ) 2> /dev/null
( timeout 500 python bench.py > $O/r02b_bench_default.json ) 2> $O/r02b_bench_default.err
ls -la $O | grep r02b
