"""Diagnostics: where does the HOST spend its time in the camera loop? (run on the GPU box)"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import bench
from g2pc.synth import make_scene, make_cameras
import gauss_render
gauss_render.PIPELINE_STREAMS = 4
dev = torch.device("cuda:0")
WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "render"          # render | render_cuda | config4 (= configs[3]: 5 M, 200 cameras)
N, NCAM, POINTS = (5_000_000, 200, 50_000_000) if WORKLOAD == "config4" else (1_000_000, 50, 10_000_000)
WORKLOAD = "render" if WORKLOAD == "config4" else WORKLOAD
scene = make_scene(N, 1237, device=dev, with_sh=(WORKLOAD == "render_cuda"))
cams = make_cameras(NCAM)
for w in range(4 if N <= 1_000_000 else 2):
    bench.one_step(scene, cams, WORKLOAD, POINTS, dev, w)
torch.cuda.synchronize()
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
bench.one_step(scene, cams, WORKLOAD, POINTS, dev, 2)
torch.cuda.synchronize()
pr.disable()
print("step wall ms", (time.perf_counter() - t) * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
