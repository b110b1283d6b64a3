#!/bin/bash
# ONE parametrised GPU-box session (replaces the per-experiment tools/gpu_round*.sh of rounds 2-3; those moved to
# tools/archive/).  Run through tools/gpurun_checked.sh so that libg2pc.so is rebuilt first:
#
#   tools/gpurun_checked.sh --timeout 1500 -- 'bash tools/gpu_session.sh <tag> <stage> [<stage> ...]'
#
# Every stage writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.  Stages:
#   tests           the whole `-m gpu` suite (+ slowest durations)          -> <tag>_pytest.log
#   tests:<expr>    `-m gpu -k <expr>`                                      -> <tag>_pytest_<expr>.log
#   smoke           __graft_entry__.smoke()                                 -> <tag>_smoke.log
#   bench           the default bench line (parity, extras, cpu baseline)   -> <tag>_bench_default.json
#   bench:<name>:<args>   `bench.py --no-parity --no-extra --no-cpu-baseline <args>`  -> <tag>_bench_<name>.json
#   prof:<name>:<args>    rocprofv3 --kernel-trace --stats of the same      -> <tag>_<name>_kernel_stats.csv (+ timeline for "prod")
#   pmc:<args>      separate FETCH_SIZE / WRITE_SIZE and SQ counter passes  -> <tag>_pmc_traffic.json, <tag>_pmc_sq.json
#   clocks:<args>   per-wave clocks of the blend (tools/chunk_work.py)      -> <tag>_chunk_clocks.txt
#   cmd:<shell>     anything else, output to <tag>_cmd<N>.log
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out
mkdir -p "$O"
T=${1:?tag}; shift
BARE="--no-parity --no-extra --no-cpu-baseline --no-profile-pass"
n=0
for st in "$@"; do
  kind=${st%%:*}; rest=${st#*:}; [ "$rest" = "$st" ] && rest=""
  case $kind in
    tests)
      if [ -z "$rest" ]; then ( timeout 1200 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -24 ) > $O/${T}_pytest.log 2>&1; tail -4 $O/${T}_pytest.log
      else f=$O/${T}_pytest_$(echo "$rest" | tr -c 'A-Za-z0-9_\n' '_').log; ( timeout 1200 python -m pytest tests -m gpu -q -x -s -k "$rest" 2>&1 | tail -60 ) > $f 2>&1; tail -3 $f; fi ;;
    smoke) ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/${T}_smoke.log 2>&1; cat $O/${T}_smoke.log ;;
    bench)
      if [ -z "$rest" ]; then ( timeout 900 python bench.py > $O/${T}_bench_default.json ) 2> $O/${T}_bench_default.err; name=default
      else name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
        ( timeout 600 python bench.py $BARE $args > $O/${T}_bench_$name.json ) 2> $O/${T}_bench_$name.err; fi
      python - <<PY
import json
try:
    d = json.load(open("$O/${T}_bench_$name.json")); print("%-28s %.3e pts/s %.3f ms" % ("$name", d["value"], d["ms_per_step"]))
except Exception as e:
    print("$name", str(e)[:120]); print(open("$O/${T}_bench_$name.err").read()[-1500:])
PY
      ;;
    prof)
      name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
      rm -rf /tmp/prof_$name
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- python $R/bench.py $BARE --steps 5 --warmup 2 $args > $O/${T}_bench_under_rocprof_$name.json ) 2> /dev/null
      db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
      if [ -n "$db" ]; then python tools/rocprof_summary.py $db > $O/${T}_${name}_kernel_stats.csv; head -6 $O/${T}_${name}_kernel_stats.csv | cut -c1-120
        [ "$name" = "prod" ] && python tools/timeline.py $db 25 > $O/${T}_timeline_prod.txt 2>&1; fi ;;
    pmc)
      # counters in their own passes, --kernel-trace only (never with a sys / hip trace); one camera per launch by default
      CMD="python $R/bench.py $BARE --steps 2 --warmup 1 ${rest:---streams 1 --camera-subset 12}"
      rm -rf /tmp/pmcF /tmp/pmcW /tmp/pmcS
      ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -o x -- $CMD > /dev/null ) 2> /dev/null
      ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -o x -- $CMD > /dev/null ) 2> /dev/null
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmcS -o x -- $CMD > /dev/null ) 2> /dev/null
      f=$(find /tmp/pmcF -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmcW -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && [ -n "$w" ] && python tools/pmc_traffic.py $f $w > $O/${T}_pmc_traffic.json
      q=$(find /tmp/pmcS -name "*counter_collection.csv" | head -1)
      [ -n "$q" ] && python tools/pmc_kernel.py $q k_blend --json > $O/${T}_pmc_sq.json
      python -c "
import json
t=json.load(open('$O/${T}_pmc_traffic.json')); s=json.load(open('$O/${T}_pmc_sq.json'))
for k in t:
    if 'blend' in k: print(k, {a: round(b/1e6,1) if isinstance(b,float) else b for a,b in t[k].items()})
print(s)" ;;
    clocks) ( timeout 600 python tools/chunk_work.py $rest ) > $O/${T}_chunk_clocks.txt 2>&1; tail -12 $O/${T}_chunk_clocks.txt ;;
    cmd) n=$((n+1)); ( timeout 900 bash -c "$rest" ) > $O/${T}_cmd$n.log 2>&1; tail -15 $O/${T}_cmd$n.log ;;
    *) echo "unknown stage $st" ;;
  esac
done
