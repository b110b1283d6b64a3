#!/bin/bash
# Experiment (GPU box): does the device context built by the FIRST full-size job of a process still render slower once the
# process warm-up (g2pc/warmup.py) has run?  A: keep it (product behaviour); B: discard it after its job (round 2's
# POOL_SKIP_FIRST_JOBS); C: keep it, larger miniature in the warm-up; D: no warm-up, keep.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="python bench.py --no-parity --no-extra --no-cpu-baseline --steps 8 --warmup 2"
for rep in 1 2; do
  for tag in A B C D; do
    case $tag in
      A) env="";; B) env="G2PC_POOL_SKIP_FIRST_JOBS=1";; C) env="G2PC_WARMUP_GAUSSIANS=30000";; D) env="G2PC_NO_WARMUP=1";;
    esac
    out=$(env $env timeout 200 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step first_job %.1f ms warmup %.1f ms' % (d['ms_per_step'], d['first_job_ms'], d['process_warmup_ms']))")
    echo "$tag rep$rep: $out"
  done
done
