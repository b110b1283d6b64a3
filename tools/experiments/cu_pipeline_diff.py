"""GPU box: native-semantics path, pipelined (no read-back) vs one camera at a time: where does the running state differ?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd")]
import torch
import camera_handler
import gaussian_pointcloud_rasterization as gpr
from gauss_handler import Gaussians
from g2pc.synth import make_scene, make_cameras
DEV = "cuda:0"
sc = make_scene(120_000, 78, device=DEV)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
tr, intr = make_cameras(7)
names = sorted(tr)
def run(streams, shrink, ncam=7):
    gpr.PIPELINE_STREAMS = streams
    R = gpr.GaussianRasterizer(G.xyz, torch.zeros_like(G.xyz), G.opacities.unsqueeze(1), colors_precomp=G.colours,
                               scales=torch.exp(sc.scales), rotations=sc.rots, visible_gaussian_threshold=0.05,
                               surface_distance_std=2.0, calculate_surface_distance=True)
    for i, k in enumerate(names[:ncam]):
        R(camera_handler.get_camera("cuda", torch.tensor(tr[k]), intr[k], colour_resolution=1280), return_image=False)
        if i == 0 and streams > 1:
            R._capacity = int(R._capacity * shrink)
    R.flush(); torch.cuda.synchronize()
    return R, (R.gaussian_max_contribution.clone(), R.gaussian_total_contribution.clone(), R.gaussian_colours.clone(),
               R.gaussian_min_surface_distance.clone(), R._winner_cam.clone())
for ncam in (1, 2, 3, 7):
    _, ref = run(1, 1.0, ncam)
    _, ref2 = run(1, 1.0, ncam)
    for tag, st, sh in (("sync again", 1, 1.0), ("pipelined", 4, 1.0), ("pipelined 2 streams", 2, 1.0), ("pipelined shrink", 4, 0.7)):
        R, got = run(st, sh, ncam)
        d = [int((x != z).reshape(x.shape[0], -1).any(1).sum()) for x, z in zip(ref, got)]
        print("ncam", ncam, tag, "rerendered", R.rerendered, "rows differing [max, total, colour, surf, winner_cam]:", d, flush=True)
