"""GPU box diagnostic: stream identities, graph captures and host-side waits of the batched camera pipeline."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc.synth import make_scene, make_cameras
from g2pc.warmup import warmup
mode, batch, slots = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
gauss_render.PIPELINE_MODE, gauss_render.CAMERA_BATCH, gauss_render.PIPELINE_STREAMS = mode, batch, slots
dev = torch.device("cuda:0")
warmup(dev, ("python",))
sc = make_scene(1_000_000, 1237, device=dev)
tr, intr = make_cameras(50)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
caps, waits = [0], [0.0]
oc, orr = gauss_render.GaussHipRenderer._capture, gauss_render.GaussHipRenderer._retire
def cap(self, *a): caps[0] += 1; return oc(self, *a)
def ret(self, sl):
    t = time.perf_counter(); r = orr(self, sl); waits[0] += time.perf_counter() - t; return r
gauss_render.GaussHipRenderer._capture, gauss_render.GaussHipRenderer._retire = cap, ret
for job in range(4):
    caps[0], waits[0] = 0, 0.0
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in tr:
        R(camera_handler.get_camera("python", torch.tensor(tr[k]), intr[k], colour_resolution=1280), return_image=False)
    t1 = time.perf_counter()
    R.flush(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("job %d: issue %.2f ms, drain %.2f ms, captures %d, host waits in _retire %.2f ms, streams %s rerendered %d" % (
        job, (t1 - t0) * 1e3, (t2 - t1) * 1e3, caps[0], waits[0] * 1e3, sorted({hex(sl.stream.cuda_stream) for sl in R.slots}), R.rerendered))
    R.close()
