#!/bin/bash
# session O: bucket depth sort after the clear-kernel fix: fault probe, primitive test, A/B of the headline job
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in 150000 1000000; do echo N=$n; PROBE_SYNC=1 timeout 90 python tools/experiments/bucket_fault_probe.py $n 8 2 2>&1 | tail -3; done
echo N=1000000 async; timeout 90 python tools/experiments/bucket_fault_probe.py 1000000 50 4 2>&1 | tail -3
timeout 200 python -m pytest tests/test_gpu_core.py -k bucket -q 2>&1 | tail -2
for m in radix bucket radix bucket; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-parity --no-extra --depth-sort $m 2> gpurun_out/r02o_$m.err | tee gpurun_out/r02o_$m.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])" || tail -3 gpurun_out/r02o_$m.err
done
