#!/bin/bash
# session S: dual-list blend kernel (k_blend_py_dl) vs the packed kernel: parity at scale, A/B of the headline job
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_graph_pipeline.py tests/test_gpu_parity_scale.py -x -q 2>&1 | tail -5
for v in 1 0 1 0; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-parity --blend-variant $v 2> gpurun_out/r02s.err | tee gpurun_out/r02s_v$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['value'])" || tail -3 gpurun_out/r02s.err
done
timeout 300 python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline --blend-variant 1 --streams 1 2>> gpurun_out/r02s.err | tee gpurun_out/r02s_v1_s1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant 1 streams 1', d['ms_per_step'], d['regions_ms_per_step'], d['parity'])"
