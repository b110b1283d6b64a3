cd ${GRAFT_REPO_ROOT:-/root/repo}
B="--no-parity --no-extra --no-cpu-baseline --no-profile-pass --camera-subset 7 --points 1250000 --steps 20 --warmup 5"
for cfg in "" "--camera-batch 1" "--camera-batch 1 --streams 7" "--camera-batch 1 --streams 8" "--streams 7" "--camera-batch 1 --streams 4" ""; do
  python bench.py $B $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share7 [%s] %.3f ms' % ('$cfg', d['ms_per_step']))"
done
python - <<'PY'
import sys, os, json, torch
sys.path[:0] = ["3dgs-to-pc_amd", "."]
PY
for i in 1 2 3; do python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --workload sample --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample %.3f ms' % d['ms_per_step'])"; done
python - <<'PY'
import sys
sys.path[:0] = ["3dgs-to-pc_amd", "."]
from g2pc import ops
ops.ONE_CALL_TAIL = False
sys.argv = ["bench.py", "--no-parity", "--no-extra", "--no-cpu-baseline", "--no-profile-pass", "--workload", "sample", "--steps", "20", "--warmup", "5"]
import bench, io, contextlib, json
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
print("sample (separate calls) %.3f ms" % json.loads(buf.getvalue().strip().splitlines()[-1])["ms_per_step"])
PY
python -m pytest tests -m gpu -q -x -k "sampler or sample or parity_scale or core or io" 2>&1 | tail -3
