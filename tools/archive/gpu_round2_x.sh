#!/bin/bash
# session X: packed (tile << shift | Gaussian) instances, keys-only tile sort: GPU tests of both semantics + bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_graph_pipeline.py tests/test_gpu_cuda_semantics.py tests/test_gpu_parity_scale.py tests/test_gpu_core.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-parity --no-extra --no-cpu-baseline 2>/dev/null | tee $O/r02x_default_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['value'])"
done
for i in 1 2; do
  timeout 200 python bench.py --steps 5 --warmup 2 --no-parity --no-extra --no-cpu-baseline --workload render_cuda 2>/dev/null | tee $O/r02x_render_cuda_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('render_cuda', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --steps 4 --warmup 2 --no-parity --no-extra --no-cpu-baseline --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default streams 1', d['ms_per_step'], d['regions_ms_per_step'])"
