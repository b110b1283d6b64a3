"""A/B aid: run bench.py against another build of the library (same box, same session).  build the other version with the Makefile's flags into a second .so in the tree (it travels with the snapshot), then alternate
`python tools/experiments/ab_lib.py <lib.so> <bench args>` and `python bench.py <bench args>` in ONE gpurun call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
from g2pc import _native as nv
nv.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
