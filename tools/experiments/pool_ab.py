"""Experiment (GPU box): bench job with and without the render-context pool, same box."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import bench, gauss_render
from g2pc.synth import make_scene, make_cameras
dev = torch.device("cuda:0")
scene = make_scene(1_000_000, 1237, device=dev)
cams = make_cameras(50)
from g2pc import _native as nv
CLEAR = False
def run(label, reps=int(os.environ.get('POOL_AB_REPS', '6'))):
    ts = []
    for r in range(reps):
        if CLEAR and nv.PROFILE is not None: nv.PROFILE.clear()
        torch.cuda.synchronize(); t = time.perf_counter()
        bench.one_step(scene, cams, "render", 10_000_000, dev, r)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize(); t = time.perf_counter()
    for r in range(reps):
        bench.one_step(scene, cams, "render", 10_000_000, dev, r)
    torch.cuda.synchronize()
    if os.environ.get("POOL_AB_BRIEF"):
        print(label, ["%.1f" % x for x in ts]); continue_brief = True
    else:
        continue_brief = False
    if continue_brief:
        return
    print(label, ["%.1f" % x for x in ts], "back-to-back: %.1f ms/step" % ((time.perf_counter() - t) * 1e3 / reps))
for mode in sys.argv[1:]:
    if mode == "pool":
        gauss_render.CONTEXT_POOL_SIZE = 2
    elif mode == "nopool":
        gauss_render.CONTEXT_POOL_SIZE = 0
        for c in gauss_render._CONTEXT_POOL: c.release()
        gauss_render._CONTEXT_POOL.clear()
    elif mode == "profile_on":
        nv.PROFILE = {}; continue
    elif mode == "profile_off":
        nv.PROFILE = None; continue
    elif mode == "clear_on":
        CLEAR = True; continue
    elif mode == "first_alone_off":
        gauss_render.FIRST_CAMERA_ALONE = False; continue
    elif mode.startswith("skip"):
        gauss_render._EXPERIMENT_SKIP = int(mode[4:]); continue      # (the G2PC_POOL_SKIP_FIRST_JOBS knob)
    run(mode)
