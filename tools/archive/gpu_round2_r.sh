#!/bin/bash
# session R: chunk-level cull at 64- / 128- / 256-pixel chunk granularity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sb in 2 1 4 1 2; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-parity --blend-subblocks $sb 2> gpurun_out/r02r.err | tee gpurun_out/r02r_sb$sb.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('subblocks $sb', d['ms_per_step'], d['value'])" || tail -3 gpurun_out/r02r.err
done
timeout 300 python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-parity --blend-subblocks 1 --streams 1 2>> gpurun_out/r02r.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('subblocks 1 streams 1', d['ms_per_step'], d['regions_ms_per_step'])"
