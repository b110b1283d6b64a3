"""GPU box: what a job costs when the image size has a non-uniform size-driven quad-tree (tile_force: every camera takes the
host-driven child passes) against the neighbouring uniform size.  usage: python tools/experiments/forced_size_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import bench
from g2pc.synth import make_scene, make_cameras
dev = torch.device("cuda:0")
scene = make_scene(1_000_000, 1237, device=dev)
for (w, h, f) in ((960, 540, 825.0), (961, 540, 826.0), (1920, 1080, 1650.0), (1936, 1089, 1664.0)):
    cams = make_cameras(50, width=w, height=h, focal=f)
    s = bench.settings("render", 10_000_000, dev)
    s = s._replace(colour_resolution=w) if hasattr(s, "_replace") else s
    try:
        s.colour_resolution = w
    except Exception:
        pass
    from gauss_handler import Gaussians
    from gauss_to_pc import convert_gaussians_to_pc
    ts = []
    for rep in range(4):
        g = Gaussians(scene.xyz, scene.scales, scene.rots, scene.colours.clone(), scene.opacities)
        torch.cuda.synchronize(); t = time.perf_counter()
        cloud, _ = convert_gaussians_to_pc(g, cams[0], cams[1], None, s, seed=rep)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("%dx%d: job ms %s  points %d" % (w, h, ["%.1f" % x for x in ts], cloud.points.shape[0]), flush=True)
