"""Diagnostics: distribution of blend work per chunk for a few cameras of the bench scene (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd")]
import numpy as np, torch
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras
dev = "cuda:0"
sc = make_scene(1_000_000, 1237, device=dev)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
tr, intr = make_cameras(50)
R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
names = sorted(tr)[:10]
for name in names:
    cam = camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name], colour_resolution=1280)
    lay = R._layout(cam.image_width, cam.image_height)
    nchunks = lay.c.num_chunks
    buf = torch.zeros(2 * nchunks, dtype=torch.int32, device=dev)
    nv.lib().g2pc_raster_debug_chunk_work(nv.ptr(buf))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nv.PROFILE = {}
    e0.record(); R(cam, return_image=False); e1.record(); torch.cuda.synchronize()
    prof = {k: round(v[1], 3) for k, v in nv.profile_summary().items()}
    nv.PROFILE = None
    w = buf.cpu().numpy().reshape(-1, 2)
    ln, done = w[:, 0], w[:, 1]
    q = lambda a: [int(np.percentile(a, p)) for p in (50, 90, 99, 99.9, 100)]
    full = (done >= ln) & (ln > 0)
    print(name, "ms %.2f" % e0.elapsed_time(e1), "chunks", nchunks, "list len p50/90/99/99.9/max", q(ln), "walked", q(done),
          "sum walked %.3g" % done.sum(), "walked-full chunks", int(full.sum()), "of which len>4096:", int((full & (ln > 4096)).sum()),
          "max walked", int(done.max()), prof)
nv.lib().g2pc_raster_debug_chunk_work(None)
