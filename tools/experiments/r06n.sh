# RECORD of a reverted experiment (profiles/r06n_lane_emit_ab.log): the 'rowemit' variant was built with -DG2PC_SAMPLER_LANE_EMIT=0 from a
# sampler.hip that held the Gaussian-centric emission kernel; that kernel and its switch were removed after the measurement.
cd ${GRAFT_REPO_ROOT:-/root/repo}
# lane emission (default) vs every row through the row-balanced kernel (rowemit), alternating, sampler workload and the 50-camera job
bash tools/experiments/ab_round.sh "--workload sample --steps 50 --warmup 10" . rowemit . rowemit . rowemit
bash tools/experiments/ab_round.sh "--steps 20 --warmup 5" . rowemit . rowemit
python -m pytest tests -m gpu -q -x -k "sampl" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06n_prof -o s -- python $GRAFT_REPO_ROOT/bench.py --workload sample --steps 20 --warmup 5 --no-parity --no-extra --no-cpu-baseline --no-profile-pass > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/r06n_prof/*kernel_stats.csv gpurun_out/r06n_prof/*/*kernel_stats.csv 2>/dev/null | head -1); head -14 "$f" | cut -c1-150
