cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/experiments/ab_round.sh "--workload render_cuda --steps 10 --warmup 3" . v109 . v109 . v109
python -m pytest tests -m gpu -q -x -k "cuda or c_entry" 2>&1 | tail -3
