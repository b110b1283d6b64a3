#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC valu_rates.hip -o libubench.so
