cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
bash tools/experiments/ab_round.sh "--workload render_cuda --steps 10 --warmup 3" . nofold . nofold . nofold
echo "--- sample chain"
rm -rf /tmp/prof_s; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o x -- python $OLDPWD/bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --workload sample --steps 5 --warmup 2 > /dev/null ) 2>/dev/null
db=$(find /tmp/prof_s -name "*_results.db" | head -1)
python tools/job_chain.py $db k_build_cov 60
echo "--- share7"
python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --camera-subset 7 --points 1250000 --steps 20 --warmup 5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share7', d['ms_per_step'])"
python bench.py --no-parity --no-extra --no-cpu-baseline --no-profile-pass --camera-subset 7 --points 1250000 --steps 20 --warmup 5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share7', d['ms_per_step'])"
