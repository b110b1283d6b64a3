#!/bin/bash
# Round 3, blend scheduling (k_blend_plan / hand-over / priorities): correctness on the MI355X, per-wave clocks of a lone
# launch with and without the new scheduling, and a sweep of the knobs through bench.py (one process per configuration).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03w}
( timeout 500 python -m pytest tests/test_gpu_graph_pipeline.py tests/test_gpu_render.py -m gpu -q -x 2>&1 | tail -15 ) > $O/${T}_pytest_blend.log 2>&1
cat $O/${T}_pytest_blend.log
# per-wave clocks: old scheduling (static order, no hand-over, r03 priority rule) vs the new defaults, 4 cameras each
( CHUNK_WORK_PIPELINE=1 CHUNK_WORK_CAMERAS=4 G2PC_BLEND_LPT=0 G2PC_BLEND_SPLIT=0 G2PC_BLEND_PRIO=0 CHUNK_WORK_OUT=${T}_clocks_old.json timeout 300 python tools/chunk_work.py 2>&1 | grep -v "^{" | tail -6 ) > $O/${T}_clocks_old.txt 2>&1
( CHUNK_WORK_PIPELINE=1 CHUNK_WORK_CAMERAS=4 CHUNK_WORK_OUT=${T}_clocks_new.json timeout 300 python tools/chunk_work.py 2>&1 | grep -v "^{" | tail -6 ) > $O/${T}_clocks_new.txt 2>&1
( CHUNK_WORK_PIPELINE=1 CHUNK_WORK_CAMERAS=4 G2PC_BLEND_SPLIT=8 G2PC_BLEND_PRIO=4 CHUNK_WORK_OUT=${T}_clocks_s8.json timeout 300 python tools/chunk_work.py 2>&1 | grep -v "^{" | tail -6 ) > $O/${T}_clocks_s8.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/*_clocks_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, e); continue
    for r in d:
        print(f.split("/")[-1], r["camera"], "span", r["span_us"], "sum_wave", r["sum_wave_us"], "exported", r.get("exported_chunks"),
              "quarters", r.get("quarters_run"), "own p50/90/99/max", r.get("own_walk_us p50/90/99/max"), "blend_ms", r["blend_region_ms"])
PY
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 10 --warmup 3"
run() {   # name, env...
  name=$1; shift
  ( env "$@" timeout 200 $B > $O/${T}_bench_$name.json ) 2> $O/${T}_bench_$name.err
  python -c "
import json
try:
    d=json.load(open('$O/${T}_bench_$name.json')); print('%-28s %.3f ms/job  %.3e pts/s' % ('$name', d['ms_per_step'], d['value']))
except Exception as e: print('$name', 'FAILED', str(e)[:100])
"
}
run old        G2PC_BLEND_LPT=0 G2PC_BLEND_SPLIT=0 G2PC_BLEND_PRIO=0
run default    G2PC_DUMMY=1
run lpt_only   G2PC_BLEND_SPLIT=0 G2PC_BLEND_PRIO=0
run lpt_prio6  G2PC_BLEND_SPLIT=0 G2PC_BLEND_PRIO=6
run split8     G2PC_BLEND_SPLIT=8 G2PC_BLEND_PRIO=4
run split16    G2PC_BLEND_SPLIT=16 G2PC_BLEND_PRIO=8
run split24    G2PC_BLEND_SPLIT=24 G2PC_BLEND_PRIO=8
run split12np  G2PC_BLEND_SPLIT=12 G2PC_BLEND_PRIO=0
run split12_nolpt G2PC_BLEND_LPT=0
run old2       G2PC_BLEND_LPT=0 G2PC_BLEND_SPLIT=0 G2PC_BLEND_PRIO=0
run default2   G2PC_DUMMY=1
