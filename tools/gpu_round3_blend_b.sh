#!/bin/bash
# Round 3, blend hand-over, second session: per-pixel debug of the hand-over against the two-call path, then clocks + sweep
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03x}
( SPLIT=1 timeout 300 python tools/experiments/debug_handover.py 2>&1 | tail -60 ) > $O/${T}_debug_split1.txt 2>&1
cat $O/${T}_debug_split1.txt
( CHUNK_WORK_PIPELINE=1 CHUNK_WORK_CAMERAS=4 G2PC_BLEND_SPLIT=24 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=8 CHUNK_WORK_OUT=${T}_clocks_s24.json timeout 300 python tools/chunk_work.py 2>&1 | grep -v "^{" | tail -6 ) > $O/${T}_clocks_s24.txt 2>&1
( CHUNK_WORK_PIPELINE=1 CHUNK_WORK_CAMERAS=4 G2PC_BLEND_SPLIT=12 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=6 CHUNK_WORK_OUT=${T}_clocks_s12.json timeout 300 python tools/chunk_work.py 2>&1 | grep -v "^{" | tail -6 ) > $O/${T}_clocks_s12.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03x_clocks_*.json")):
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
run s24_p8     G2PC_BLEND_SPLIT=24 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=8
run s16_p8     G2PC_BLEND_SPLIT=16 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=8
run s32_p8     G2PC_BLEND_SPLIT=32 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=8
run s12_p6     G2PC_BLEND_SPLIT=12 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=6
run s24_p0     G2PC_BLEND_SPLIT=24 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=0
run s24_nolpt  G2PC_BLEND_LPT=0 G2PC_BLEND_SPLIT=24 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=8
run s20_b1     G2PC_BLEND_SPLIT=20 G2PC_BLEND_MINLEFT=256 G2PC_BLEND_PRIO=8
