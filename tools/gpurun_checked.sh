#!/bin/bash
# Builds libg2pc.so (and the oracle's C restatement) from the current sources before shipping the tree to the GPU box:
# the .so travels with the snapshot, a stale one silently runs old kernels.  Usage: tools/gpurun_checked.sh [--timeout S] -- 'cmd'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun "$@"
