#!/bin/bash
# session Q: chunk-level cull in the packed blend: parity at scale, render tests, bench (default + exact + streams 1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_graph_pipeline.py tests/test_gpu_parity_scale.py -x -q 2>&1 | tail -5
for a in "" "--streams 1" "--t-floor 0"; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline $a 2> gpurun_out/r02q.err | tee "gpurun_out/r02q_bench$(echo $a | tr -d ' -').json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['value'], d.get('parity'), d['regions_ms_per_step'])" || tail -3 gpurun_out/r02q.err
done
