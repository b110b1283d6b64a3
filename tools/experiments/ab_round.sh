#!/bin/bash
# A/B aid: one bench line per variant tag, in the order given (repeat tags to see the box's noise):
#   tools/experiments/ab_round.sh "<bench args>" base . f . r8      ("." = the default library libg2pc.so)
# variants are built by tools/experiments/build_variant.sh into 3dgs-to-pc_amd/g2pc/libg2pc_<tag>.so
cd "$(dirname "$0")/../.."
args=$1; shift
BARE="--no-parity --no-extra --no-cpu-baseline --no-profile-pass"
for v in "$@"; do
  if [ "$v" = "." ]; then cmd="python bench.py"; else cmd="python tools/experiments/ab_lib.py 3dgs-to-pc_amd/g2pc/libg2pc_$v.so"; fi
  out=$(timeout 300 $cmd $BARE $args 2>/dev/null | tail -1)
  python - "$v" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print("%-8s %.3f ms  %.3e pts/s" % (sys.argv[1], d["ms_per_step"], d["value"]))
except Exception as e:
    print(sys.argv[1], "failed", sys.argv[2][-200:])
PY
done
