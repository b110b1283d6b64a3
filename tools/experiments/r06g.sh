cd ${GRAFT_REPO_ROOT:-/root/repo}
# "." = plane-major SH (default); rows = the reference's layout read as 16-byte vectors (SH_PLANES off); old = round 5's scalar loads
cat > /tmp/rows.py <<'PY'
import sys
sys.path[:0] = ["3dgs-to-pc_amd", "."]
import gaussian_pointcloud_rasterization as gpr
gpr.SH_PLANES = False
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
PY
B="--no-parity --no-extra --no-cpu-baseline --no-profile-pass --workload render_cuda --steps 10 --warmup 3"
for i in 1 2 3; do
  python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planes %.3f ms' % d['ms_per_step'])"
  python /tmp/rows.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows   %.3f ms' % d['ms_per_step'])"
  python tools/experiments/ab_lib.py 3dgs-to-pc_amd/g2pc/libg2pc_old.so $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old(planes ignored: scalar loads of planes? no) %.3f ms' % d['ms_per_step'])"
done
python -m pytest tests -m gpu -q -x -k "cuda or c_entry or million" 2>&1 | tail -3
