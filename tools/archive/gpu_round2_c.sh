#!/bin/bash
# Round-2 GPU session C: VALU / LDS issue-rate micro-benchmark, parity test, SQ counters of the blend kernel.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 120 python tools/experiments/ubench/run_valu_rates.py > $O/r02c_valu_rates.json ) 2> $O/r02c_valu_rates.err
( timeout 400 python -m pytest tests/test_gpu_parity_scale.py -m gpu -q -s 2>&1 | tail -60 ) > $O/r02c_pytest_parity.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 2 --warmup 1 --streams 1 --camera-subset 10"
cd /tmp
rm -rf /tmp/pmc1 /tmp/pmc2
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc1 -o x -- $CMD > /dev/null ) 2> $GRAFT_REPO_ROOT/$O/r02c_pmc1.err
( timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pmc2 -o x -- $CMD > /dev/null ) 2> $GRAFT_REPO_ROOT/$O/r02c_pmc2.err
for i in 1 2; do
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $f k_blend --json > $GRAFT_REPO_ROOT/$O/r02c_pmc_sq_$i.json
done
cd $GRAFT_REPO_ROOT
ls -la $O | grep r02c
