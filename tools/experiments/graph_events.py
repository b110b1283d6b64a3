"""Experiment (GPU box): are HIP events recorded inside a captured graph usable for timing?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras
dev = "cuda:0"
sc = make_scene(200_000, 1237, device=dev)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
tr, intr = make_cameras(12)
R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
nv.PROFILE = {}
for name in sorted(tr):
    cam = camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name], colour_resolution=1280)
    R(cam, return_image=False)
R.flush()
torch.cuda.synchronize()
print("events_refused", R.events_refused, "last error:", nv.lib().g2pc_last_error())
print({k: (len(v), [round(x, 3) if isinstance(x, float) else "ev" for x in v][:12]) for k, v in nv.PROFILE.items()})
print("slots with events", [bool(sl.events) for sl in R.slots], "rerendered", R.rerendered)
