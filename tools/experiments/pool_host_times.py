"""Experiment (GPU box): where does the host spend the camera loop in the slow (first context) and fast states?"""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import bench, gauss_render
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras
dev = torch.device("cuda:0")
scene = make_scene(1_000_000, 1237, device=dev)
cams = make_cameras(50)
T = collections.OrderedDict()
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + (time.perf_counter() - t) * 1e3; return r
    return w
L = nv.lib()
for f in ("g2pc_graph_launch", "g2pc_raster_camera_update_py"):
    orig = getattr(L, f); setattr(L, f, timed(f, orig))
R = gauss_render.GaussHipRenderer
R._retire = timed("_retire (wait for the slot)", R._retire)
R._camera_struct = timed("_camera_struct", R._camera_struct)
R._render_pipelined = timed("_render_pipelined total", R._render_pipelined)
def run(label, reps=5):
    for r in range(reps):
        T.clear()
        torch.cuda.synchronize(); t = time.perf_counter()
        bench.one_step(scene, cams, "render", 10_000_000, dev, r)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    print(label, "step %.1f ms" % dt, {k: round(v, 2) for k, v in T.items()})
for mode in sys.argv[1:]:
    if mode == "nopool":
        gauss_render.CONTEXT_POOL_SIZE = 0
        for c in gauss_render._CONTEXT_POOL: c.release()
        gauss_render._CONTEXT_POOL.clear()
    else:
        gauss_render.CONTEXT_POOL_SIZE = 2
    run(mode)
