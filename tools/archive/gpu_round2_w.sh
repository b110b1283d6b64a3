#!/bin/bash
# session W: PMC passes of the final blend kernel (HBM traffic: FETCH_SIZE / WRITE_SIZE separately; SQ counters), --streams 1
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/bench.py --no-parity --no-extra --no-cpu-baseline --steps 2 --warmup 1"
cd /tmp; rm -rf /tmp/pmcF /tmp/pmcW /tmp/pmcS
( timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -o x -- $CMD > /dev/null ) 2> /dev/null
( timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -o x -- $CMD > /dev/null ) 2> /dev/null
f=$(find /tmp/pmcF -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmcW -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && [ -n "$w" ] && python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $f $w > $O/r02w_pmc_traffic.json
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmcS -o x -- $CMD --streams 1 --camera-subset 10 > /dev/null ) 2> /dev/null
f=$(find /tmp/pmcS -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $f k_ --json > $O/r02w_pmc_sq.json
python - <<PY
import json
t=json.load(open("$O/r02w_pmc_traffic.json")); s=json.load(open("$O/r02w_pmc_sq.json"))
for k in t:
    if "blend" in k or "preprocess" in k or "bk_sort" in k: print(k, t[k])
for k in s:
    if "blend" in k: print(k, s[k])
PY
