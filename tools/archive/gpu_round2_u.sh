#!/bin/bash
# session U: SQ counters of the dual-list blend kernel (two passes, --streams 1)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 2 --warmup 1 --streams 1 --camera-subset 10"
cd /tmp; rm -rf /tmp/pmc1 /tmp/pmc2
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc1 -o x -- $CMD > /dev/null ) 2> $GRAFT_REPO_ROOT/$O/r02u_pmc1.err
( timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d /tmp/pmc2 -o x -- $CMD > /dev/null ) 2> $GRAFT_REPO_ROOT/$O/r02u_pmc2.err
for i in 1 2; do
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $f k_blend --json > $GRAFT_REPO_ROOT/$O/r02u_pmc_sq_blend_pass$i.json
done
cat $GRAFT_REPO_ROOT/$O/r02u_pmc_sq_blend_pass1.json $GRAFT_REPO_ROOT/$O/r02u_pmc_sq_blend_pass2.json; tail -3 $GRAFT_REPO_ROOT/$O/r02u_pmc2.err
