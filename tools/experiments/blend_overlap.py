"""Experiment (GPU box): can two blend launches (different cameras, different streams) overlap?  Times k blends issued
serially on one stream against the same k blends issued on k streams."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import ctypes as C
import torch
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras
dev = torch.device("cuda:0")
sc = make_scene(1_000_000, 1237, device=dev)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
tr, intr = make_cameras(50)
names = sorted(tr)
R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
# warm the state with 8 cameras (so that visibility updates are as rare as in steady state)
for name in names[:8]:
    R(camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name], colour_resolution=1280), return_image=False)
R.flush(); torch.cuda.synchronize()
K = 4
cams = [camera_handler.get_camera("python", torch.tensor(tr[n]), intr[n], colour_resolution=1280) for n in names[8:8 + K]]
lay = R._layout(1280, 720)
slots = R.slots[:K]
cap = R.capacity
def prep(sl, cam, slot):
    R._camera_struct(cam, sl.job.cam); sl.job.camera_slot, sl.job.t_floor = slot, R.t_floor
    nv.check(R._camera_call(sl, lay, cap, 1), "front+bin")
def blend(sl):
    nv.check(R._camera_call(sl, lay, cap, 2), "blend")
for i, (sl, cam) in enumerate(zip(slots, cams)):
    prep(sl, cam, 20 + i)
torch.cuda.synchronize()
def timeit(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best
one = [timeit(lambda sl=sl: blend(sl)) for sl in slots]
print("single blends (ms):", [round(x, 3) for x in one], "sum %.3f" % sum(one))
print("%d blends on %d streams at once: %.3f ms" % (K, K, timeit(lambda: [blend(sl) for sl in slots])))
print("2 blends on 2 streams at once: %.3f ms (singles %.3f + %.3f)" % (timeit(lambda: [blend(sl) for sl in slots[:2]]), one[0], one[1]))
# the front+bin chain alone, and against a running blend
print("front+bin chain alone: %.3f ms" % timeit(lambda: prep(slots[0], cams[0], 20)))
print("front+bin (stream 0) with a blend running on stream 1: %.3f ms (blend alone %.3f)" % (timeit(lambda: (blend(slots[1]), prep(slots[0], cams[0], 20))), one[1]))
print("2 front+bin chains on 2 streams: %.3f ms" % timeit(lambda: (prep(slots[0], cams[0], 20), prep(slots[1], cams[1], 21))))
