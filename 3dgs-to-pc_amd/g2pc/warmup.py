"""
Process warm-up of the hot path.

The first launch of a kernel loads its code object, the first use of a torch operator loads torch's, the first hipMalloc
of a size class pays the driver, and a kernel launched for the first time INSIDE a stream capture leaves a graph that
replays ~25 % slower for as long as it lives.  A one-shot `python gauss_to_pc.py ...` used to pay all of that inside its
only job (measured on MI355X, profiles/r03h_first_job_stages.txt: 263 ms for the first 1 M-Gaussian / 50-camera /
10 M-point job of a process against 17.7 ms for the fourth -- 94 ms in the camera loop's first launches, 128 ms in the first
use of the torch operators behind the getters and the filter, 18 ms of first allocations).

`warmup(device)` runs a MINIATURE of the whole pipeline -- 512 Gaussians, 6 cameras of 64 x 48 through the captured-graph
path of the python-semantics renderer and 2 through the native-rasteriser semantics, cull, filter, validate, magnitudes,
allocation, sampling -- once per process and device: every kernel of libg2pc.so and every torch operator the pipeline
touches is loaded, outside any real job.  The CLI starts it on a thread while the host parses the input files
(gauss_to_pc.main); bench.py calls it with the library load, before the first job it times (`first_job_ms`).
"""
import threading

import torch

_DONE = {}
_LOCK = threading.Lock()


def warmup(device, semantics=("python", "cuda")):
    """Idempotent per (process, device).  Returns the seconds it took (0.0 when already warm).

    The warm-up is an optimisation, never a reason to fail a conversion: whatever it raises is logged and swallowed (the
    real job then simply pays the first-use costs itself).  It runs with `device` as the CURRENT device of the calling
    thread -- torch.cuda.set_device is per thread, and a daemon thread of rank r > 0 would otherwise pin buffers, create
    events and launch libg2pc's kernels (which take the current device) on GPU 0 -- and as a single-process job: under
    torch.distributed the miniature must not issue collectives from a side thread while the ranks are busy elsewhere."""
    import time
    device = torch.device(device)
    key = str(device)
    with _LOCK:
        if _DONE.get(key):
            return 0.0
        t0 = time.perf_counter()
        try:
            if device.type == "cuda":
                with torch.cuda.device(device):
                    _run(device, semantics)
                    torch.cuda.synchronize(device)
            else:
                _run(device, semantics)
        except Exception as e:                              # noqa: BLE001 -- see docstring
            import sys
            import traceback
            print("g2pc warm-up failed (%s: %s); continuing without it" % (type(e).__name__, e), file=sys.stderr)
            if __import__("os").environ.get("G2PC_WARMUP_DEBUG"):
                traceback.print_exc()
            try:
                import gauss_render
                gauss_render.clear_context_pool()
            except Exception:                               # noqa: BLE001
                pass
        _DONE[key] = True
        return time.perf_counter() - t0


def warmup_in_background(device, semantics=("python", "cuda")):
    """Start the warm-up on a daemon thread (the CLI does, while it reads the .ply); join() the result before rendering."""
    t = threading.Thread(target=warmup, args=(device, semantics), daemon=True)
    t.start()
    return t


def _run(device, semantics):
    import gauss_render
    from gauss_handler import Gaussians
    from gauss_to_pc import GaussPointCloudSettings, convert_gaussians_to_pc
    from g2pc.synth import make_scene, make_cameras
    import os
    n_mini = int(os.environ.get("G2PC_WARMUP_GAUSSIANS", "512"))
    sc = make_scene(n_mini, 7, device=device, scale_lo=0.02, scale_hi=0.1)
    for sem in semantics:
        cu = sem == "cuda"
        tr, intr = make_cameras(2 if cu else gauss_render.PIPELINE_STREAMS + 2, width=64, height=48, focal=55.0)
        g = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours.clone(), sc.opacities)
        s = GaussPointCloudSettings(
            renderer_type=sem, num_points=4000, prioritise_visible_gaussians=True, mahalanobis_distance_std=2.0,
            camera_skip_rate=0, render_colours=True, min_opacity=0.0, bounding_box_min=None, bounding_box_max=None,
            calculate_normals=True, cull_large_percentage=0.0, remove_unrendered_gaussians=True, colour_resolution=64,
            max_sh_degree=3, exact_num_points=False, visibility_threshold=0.01, surface_distance_std=2.0 if cu else None,
            generate_mesh=False, quiet=True, device=str(device))
        convert_gaussians_to_pc(g, tr, intr, None, s, seed=1, keep_render_context=False, single_process=True)
    gauss_render.clear_context_pool()       # the miniature's device context is of no use to a real scene
