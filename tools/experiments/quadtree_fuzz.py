"""CPU (emulator): random image sizes, tile limits, Gaussian counts and crowding through the renderer's quad-tree path against
oracle/ref_render.py (whose queue is the reference's).  usage: python tools/experiments/quadtree_fuzz.py <seed> <cases> [pipelined] [gpu]
`pipelined` (round 4): 4 cameras per case through the capture-and-replay pipeline (floor 1e-6): static and on-demand child passes,
host levels with a child pass's sequence numbers.
Round 3: 300 cases; found the no-leaf layout crash and the children the image border clips to one pixel (kept, painted)."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3dgs-to-pc_amd'), os.path.join(ROOT, 'oracle')]
import numpy as np
from g2pc import _native as nv
GPU = "gpu" in sys.argv[3:]                   # ... gpu: the real library on cuda:0 (GPU box), oracle on the host
DEV = "cuda:0" if GPU else "cpu"
if not GPU:
    from emu_util import build_emu
    nv._inject_for_tests(build_emu())
from render_checks import run_vs_oracle
rng = np.random.default_rng(int(sys.argv[1]))
PIPE = "pipelined" in sys.argv[3:]
ONLY = next((int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("only=")), None)      # replay ONE case of the stream
if PIPE:
    import gauss_render
    gauss_render.PIPELINE_IN_EMULATOR = True
    gauss_render.BLEND_SUBBLOCKS = 2
bad = 0
for it in range(int(sys.argv[2])):
    W = int(rng.integers(40, 420)); H = int(rng.integers(30, 260))
    mt = int(rng.choice([6, 9, 14, 25, 60])); mg = int(rng.choice([15, 60, 250, 1000]))
    n = int(rng.integers(300, 2500)); crowd = float(rng.choice([0.15, 0.4, 1.0]))
    sc = (0.004, float(rng.choice([0.02, 0.06])))
    if ONLY is not None and it != ONLY:
        continue
    t = time.time()
    try:
        if PIPE:
            # the same four cameras through the two-call path and through the pipeline: the pipeline must land on the SAME
            # numbers (floor mode flips an arg-max between pixels tying to ~1e-6 now and then: identically in both)
            gauss_render.PIPELINE_IN_EMULATOR = False
            gauss_render.PIPELINE_STREAMS = 1 if GPU else gauss_render.PIPELINE_STREAMS     # (one stream = the two-call path)
            ref = run_vs_oracle(n, 1000 + it, W, H, 0.9 * W, 4, device=DEV, scale=sc, t_floor=1e-6, max_tile_size=mt, max_gaussians_per_tile=mg,
                                xyz_scale=crowd, pipelined=False)
            gauss_render.PIPELINE_IN_EMULATOR = True
            gauss_render.PIPELINE_STREAMS = 4 if GPU else gauss_render.PIPELINE_STREAMS
            res = run_vs_oracle(n, 1000 + it, W, H, 0.9 * W, 4, device=DEV, scale=sc, t_floor=1e-6, max_tile_size=mt, max_gaussians_per_tile=mg,
                                xyz_scale=crowd, pipelined=True)
            same = all(ref[k] == res[k] for k in ("contribution", "colour", "colour_off_gaussians", "flips", "colour_off_tiny"))
            res["image"] = 0.0 if same else 1.0                # (the pipeline returns no image: this slot carries "equal to the two-call path")
        else:
            res = run_vs_oracle(n, 1000 + it, W, H, 0.9 * W, 1, device=DEV, scale=sc, t_floor=0.0, max_tile_size=mt, max_gaussians_per_tile=mg, xyz_scale=crowd)
    except NotImplementedError as e:
        print(it, W, H, mt, mg, n, crowd, "NotImplemented:", str(e)[:70]); continue
    except ValueError as e:                      # (more leaves + split children than the keys' 14-bit tile field numbers)
        if "sequence numbers" not in str(e):
            raise
        print(it, W, H, mt, mg, n, crowd, "Refused:", str(e)[:70]); continue
    ok = res["image"] < 2e-5 and res["contribution"] < 2e-5 and res["flips"] == 0 and res["colour_off_gaussians"] <= (8 if PIPE else 2)
    bad += (not ok)
    print(it, W, H, mt, mg, n, crowd, "split", res["split_leaves"], "childpass", res.get("child_pass_cameras"), "host", res.get("host_driven"), "img %.1e c %.1e col %.1e off %d flips %d" % (res["image"], res["contribution"], res["colour"], res["colour_off_gaussians"], res["flips"]), "OK" if ok else "MISMATCH", "%.1fs" % (time.time() - t), flush=True)
print("mismatches", bad)
