"""Diagnostics (GPU box): wall time of every stage of one bench step, stages separated by device synchronisation
(the camera loop is timed as a whole: host issue time of the 50 calls, then the drain)."""
import sys, os, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import torch
import bench
import gauss_to_pc, gauss_render, gauss_handler, camera_handler
from g2pc.synth import make_scene, make_cameras
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--points", type=int, default=10_000_000)
ap.add_argument("--subblocks", type=int, default=None)
ap.add_argument("--workload", default="render", choices=["render", "render_cuda", "sample"])
ap.add_argument("--every-job", action="store_true", help="print the table of every job of the process (job 1 = what a one-shot conversion pays)")
ap.add_argument("--jobs", type=int, default=3)
a = ap.parse_args()
if a.subblocks:
    gauss_render.BLEND_SUBBLOCKS = a.subblocks
dev = torch.device("cuda:0")
T = collections.OrderedDict()
def timed(name, fn, sync=True):
    def w(*args, **kw):
        if sync: torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*args, **kw)
        if sync: torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    return w
G = gauss_handler.Gaussians
G.__init__ = timed("Gaussians.__init__", G.__init__)
G.calculate_normals = timed("calculate_normals", G.calculate_normals)
G.filter_gaussians = timed("filter_gaussians", G.filter_gaussians)
G.validate_covariances = timed("validate_covariances", G.validate_covariances)
gauss_to_pc.get_renderer = timed("get_renderer", gauss_to_pc.get_renderer)
gauss_to_pc.get_camera = timed("get_camera (host, 50x)", gauss_to_pc.get_camera, sync=False)
if a.workload == "render_cuda":
    import gaussian_pointcloud_rasterization as gpr
    R = gpr.GaussianRasterizer
    R.forward = timed("renderer.forward issue (host, 50x)", R.forward, sync=False)
    R.get_gaussians_with_low_surface_distance = timed("get_gaussians_with_low_surface_distance", R.get_gaussians_with_low_surface_distance)
else:
    R = gauss_render.GaussHipRenderer
    R.__call__ = timed("renderer.__call__ issue (host, 50x)", R.__call__, sync=False)
    R._capture = timed("  of which graph capture", R._capture, sync=False)
    R._render_sync = timed("  of which first camera (two-call path)", R._render_sync, sync=False)
R.get_gaussian_colours = timed("drain + get_gaussian_colours", R.get_gaussian_colours)
R.get_visible_gaussians = timed("get_visible_gaussians", R.get_visible_gaussians)
R.get_total_gaussian_contributions = timed("get_total_contributions", R.get_total_gaussian_contributions)
gauss_to_pc.generate_pointcloud = timed("generate_pointcloud", gauss_to_pc.generate_pointcloud)
scene = make_scene(a.gaussians, 1237, device=dev, with_sh=(a.workload == "render_cuda"))
cams = make_cameras(50)
def report(wall):
    print("step wall ms (with stage syncs) %.2f" % wall)
    for k, v in T.items():
        print("%-45s %8.3f ms" % (k, v))
    print("sum of top-level stages %.2f" % sum(v for k, v in T.items() if not k.startswith("  ")))
for rep in range(a.jobs):
    T.clear()
    torch.cuda.synchronize()
    t = time.perf_counter()
    bench.one_step(scene, cams if a.workload != "sample" else None, a.workload, a.points, dev, rep)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) * 1e3
    if a.every_job:
        print("---- job %d of this process" % (rep + 1))
        report(wall)
if not a.every_job:
    report(wall)
