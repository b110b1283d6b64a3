"""Diagnostics: wall time of the pipeline phases of one job (synchronised between phases; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import bench
from g2pc.synth import make_scene, make_cameras
from gauss_handler import Gaussians
import gauss_render, camera_handler, gauss_to_pc as g2p
dev = torch.device("cuda:0")
scene = make_scene(1_000_000, 1237, device=dev)
tr, intr = make_cameras(50)
s = bench.settings("render", 10_000_000, dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T()
    G = Gaussians(scene.xyz, scene.scales, scene.rots, scene.colours.clone(), scene.opacities); G.calculate_normals()
    t1 = T()
    R = gauss_render.get_renderer("python", G.xyz, torch.unsqueeze(torch.clone(G.opacities), 1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
    t2 = T()
    for name in tr:
        R(camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name], colour_resolution=1280), return_image=False)
    R.flush(); t3 = T()
    G.colours = R.get_gaussian_colours(); G.add_gaussians_to_cull(R.get_visible_gaussians())
    G.apply_min_opacity(0.0); G.apply_bounding_box(None, None)
    culled = G.filter_gaussians(); contrib = R.get_total_gaussian_contributions()[culled]
    t4 = T()
    keep = G.validate_covariances(); contrib = contrib[keep]
    t5 = T()
    pts, cols, nrm = g2p.generate_pointcloud(G, 10_000_000, contributions=contrib, quiet=True, seed=rep)
    t6 = T()
    print("ms: cov+normals %.2f  renderer init %.2f  cameras %.2f  colours+cull+filter %.2f  validate %.2f  generate_pointcloud %.2f  total %.2f  kept %d pts %d"
          % tuple([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t0, t6))] + [G.xyz.shape[0], pts.shape[0]]))
