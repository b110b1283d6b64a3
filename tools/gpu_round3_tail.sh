#!/bin/bash
# Round 3: the blend's tail kernel (k_blend_tail).  GPU tests of the pipeline, lone-launch region times with / without the
# hand-over, and a sweep of (cap_batches, min_left) through bench.py -- one process per configuration.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03zc}
( timeout 600 python -m pytest tests/test_gpu_graph_pipeline.py tests/test_gpu_render.py tests/test_gpu_parity_scale.py -m gpu -q -x 2>&1 | tail -8 ) > $O/${T}_pytest_tail.log 2>&1
cat $O/${T}_pytest_tail.log
for cfg in "0,0" "16,256" "8,256" "24,256"; do
  tag=$(echo $cfg | tr ',' '_')
  ( G2PC_BLEND_TAIL=$cfg CHUNK_WORK_PIPELINE=1 CHUNK_WORK_CAMERAS=6 CHUNK_WORK_OUT=${T}_clocks_$tag.json timeout 300 python tools/chunk_work.py 2>&1 | grep -v "^{" | tail -3 ) > $O/${T}_clocks_$tag.txt 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03zc_clocks_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], "blend region ms per camera:", [r["blend_region_ms"] for r in d], "main span us:", [r["span_us"] for r in d])
PY
B="python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --steps 10 --warmup 3"
run() {
  name=$1; shift
  ( env "$@" timeout 200 $B > $O/${T}_bench_$name.json ) 2> $O/${T}_bench_$name.err
  python -c "
import json
try:
    d=json.load(open('$O/${T}_bench_$name.json')); print('%-24s %.3f ms/job  %.3e pts/s' % ('$name', d['ms_per_step'], d['value']))
except Exception as e: print('$name', 'FAILED', str(e)[:100])
"
}
run off        G2PC_BLEND_TAIL=0,0
run c16_m256   G2PC_BLEND_TAIL=16,256
run c12_m256   G2PC_BLEND_TAIL=12,256
run c8_m256    G2PC_BLEND_TAIL=8,256
run c8_m64     G2PC_BLEND_TAIL=8,64
run c20_m256   G2PC_BLEND_TAIL=20,256
run c24_m512   G2PC_BLEND_TAIL=24,512
run c32_m512   G2PC_BLEND_TAIL=32,512
run c6_m128    G2PC_BLEND_TAIL=6,128
run off2       G2PC_BLEND_TAIL=0,0
run c16_again  G2PC_BLEND_TAIL=16,256
