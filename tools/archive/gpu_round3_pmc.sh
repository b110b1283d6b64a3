#!/bin/bash
# PMC collections of the blend kernel (one camera per launch: bench.py --streams 1), counters in their own passes with
# --kernel-trace only: FETCH_SIZE, WRITE_SIZE (HBM traffic, tools/pmc_traffic.py), SQ issue / residency counters.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r03t}
CMD="python $R/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 2 --warmup 1 --streams 1 --camera-subset 12"
rm -rf /tmp/pmcF /tmp/pmcW /tmp/pmcS
( timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -o x -- $CMD > /dev/null ) 2> /dev/null
( timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -o x -- $CMD > /dev/null ) 2> /dev/null
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmcS -o x -- $CMD > /dev/null ) 2> /dev/null
f=$(find /tmp/pmcF -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmcW -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && [ -n "$w" ] && python $R/tools/pmc_traffic.py $f $w > $R/gpurun_out/${T}_pmc_traffic.json
q=$(find /tmp/pmcS -name "*counter_collection.csv" | head -1)
[ -n "$q" ] && python $R/tools/pmc_kernel.py $q k_blend --json > $R/gpurun_out/${T}_pmc_sq.json
python -c "
import json
t=json.load(open('$R/gpurun_out/${T}_pmc_traffic.json')); s=json.load(open('$R/gpurun_out/${T}_pmc_sq.json'))
for k in t:
    if 'blend' in k or 'bk_sort' in k or 'preprocess' in k: print(k, {a: round(b/1e6,1) if isinstance(b,float) else b for a,b in t[k].items()})
print(s)"
