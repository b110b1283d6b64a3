#!/usr/bin/env python
"""
bench.py -- coloured points / second of the 3DGS -> point-cloud hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload render|sample] [--no-cpu-baseline]

A "step" is one whole pass of the hot path over one synthetic scene whose inputs (xyz, log-scales, rotations,
opacities, colours) are already resident in HBM when the timed region starts:
  render (BASELINE.json configs[2], the configuration the metric is quoted on):
      covariances+normals -> 50 cameras x (preprocess, depth sort, tile binning, blend + visibility) ->
      colours / cull / filter -> validate -> magnitudes -> distribute -> sample 10M points
  sample (configs[1]): covariances+normals -> validate -> magnitudes(opacity) -> distribute -> sample 10M points
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), launched by torch.distributed.run.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=[None, "render", "sample", "render_cuda"])
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--cameras", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also gather the sharded cloud on rank 0 inside the timed job (36 B per point over xGMI)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU brings its own 50 cameras and 10M-point budget (50N cameras, 10M*N points "
                         "on the shared scene); strong = the N = 1 job split N ways")
    ap.add_argument("--sort-bits", type=int, default=0, help="tuning aid: wide radix digit bits (8 or 11)")
    ap.add_argument("--sort-small", type=int, default=2 << 20, help="tuning aid: inputs up to this many keys use 4 keys/thread")
    ap.add_argument("--blend-subblocks", type=int, default=0, help="tuning aid: 8x8 sub-blocks per blend wave (1, 2, 4)")
    ap.add_argument("--scene-scales", type=float, nargs=2, default=None, metavar=("LO", "HI"),
                    help="diagnostic: Gaussian scale range of the synthetic scene (default 0.002 0.02 = SURVEY.md's)")
    ap.add_argument("--no-context-pool", action="store_true", help="tuning aid: every job captures its camera graphs anew")
    ap.add_argument("--streams", type=int, default=0, help="tuning aid: cameras in flight (HIP streams) of the renderer")
    ap.add_argument("--camera-subset", type=int, default=0, help="profiling aid: render only the first k cameras of the rig")
    ap.add_argument("--t-floor", type=float, default=None, help="blend transmittance floor (default: gauss_render.DEFAULT_T_FLOOR)")
    return ap.parse_args()


def have_renderer():
    try:
        import gauss_render
        return hasattr(gauss_render, "GaussHipRenderer")
    except Exception:
        return False


def settings(workload, num_points, device):
    from gauss_to_pc import GaussPointCloudSettings
    cu = workload == "render_cuda"        # configs[4]: native-rasteriser semantics + surface cull + exact points + SH
    return GaussPointCloudSettings(
        renderer_type="cuda" if cu else "python", num_points=num_points, prioritise_visible_gaussians=True,
        mahalanobis_distance_std=2.0, camera_skip_rate=0, render_colours=(workload != "sample"), min_opacity=0.0,
        bounding_box_min=None, bounding_box_max=None, calculate_normals=True, cull_large_percentage=0.0,
        remove_unrendered_gaussians=True, colour_resolution=1280, max_sh_degree=3, exact_num_points=cu,
        visibility_threshold=0.05, surface_distance_std=2.0 if cu else None, generate_mesh=False, quiet=True,
        device=str(device))


GATHER_OUTPUT = False      # N > 1: the cloud stays sharded by Gaussian index (one PLY part per rank) unless --gather


def one_step(scene, cams, workload, num_points, device, seed):
    """One pass of the hot path; returns the number of coloured points produced."""
    from gauss_handler import Gaussians
    from gauss_to_pc import convert_gaussians_to_pc
    g = Gaussians(scene.xyz, scene.scales, scene.rots, scene.colours.clone(), scene.opacities, shs=scene.shs)
    transforms, intr = cams if cams is not None else (None, None)
    cloud, _ = convert_gaussians_to_pc(g, transforms, intr, None, settings(workload, num_points, device), seed=seed,
                                       render_shs=(workload == "render_cuda"))
    from g2pc.dist import gather_pointcloud, rank_world
    n_local = cloud.points.shape[0]
    if GATHER_OUTPUT and rank_world()[1] > 1:
        gather_pointcloud(cloud, dst=0)            # --gather: assemble the cloud on rank 0 inside the timed job
    return n_local


def algorithmic_bytes(workload, n, n_kept, m, cams, stats):
    """B_geom = 116 N; B_samp = 56 N_kept + 36 M; B_cam = 156 N + (76 + 24 p) L + 32 W H."""
    b = 116.0 * n + 56.0 * n_kept + 36.0 * m
    if workload == "render":
        for (L, p, wh) in stats:
            b += 156.0 * n + (76.0 + 24.0 * p) * L + 32.0 * wh
    return b


def pmc_traffic(region, a):
    """HBM bytes per launch of the region's kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    in separate runs of THIS bench command, tools/pmc_traffic.py -> profiles/r01_c_pmc_traffic.json); only valid for
    the default workload it was collected on.  Corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE
    counts 16-byte-per-lane loads (what the blend's record gathers are) at half their bytes, so it is doubled;
    WRITE_SIZE is taken as is; both are KB."""
    import gauss_render
    from g2pc import tiles
    sub = gauss_render.BLEND_SUBBLOCKS or tiles.SUBBLOCKS_PER_CHUNK
    kernel = {"raster_blend": {1: "void g2pc::k_blend_py<1, 4>", 2: "void g2pc::k_blend_py_pk<4>"}.get(sub),
              "sampler_emit": "g2pc::k_emit_wave"}.get(region)
    path = os.path.join(ROOT, "profiles", "r01_f_pmc_traffic.json" if sub == 2 else "r01_c_pmc_traffic.json")
    if (kernel is None or not os.path.isfile(path) or (a.gaussians, a.cameras) != (1_000_000, 50)
            or (region == "raster_blend" and gauss_render.DEFAULT_T_FLOOR != 1e-6)):
        return None
    rec = json.load(open(path)).get(kernel)
    return rec["hbm_bytes_fetch_x2"] if rec else None


def pmc_valu(region, a):
    """VALU wave-instructions per launch of the blend kernel from the committed SQ counter pass (tools/pmc_kernel.py ->
    profiles/r01_f_pmc_sq.json); same validity conditions as pmc_traffic."""
    import gauss_render
    from g2pc import tiles
    sub = gauss_render.BLEND_SUBBLOCKS or tiles.SUBBLOCKS_PER_CHUNK
    path = os.path.join(ROOT, "profiles", "r01_f_pmc_sq.json")
    if (region != "raster_blend" or sub != 2 or not os.path.isfile(path) or (a.gaussians, a.cameras) != (1_000_000, 50)
            or gauss_render.DEFAULT_T_FLOOR != 1e-6):
        return None
    rec = json.load(open(path)).get("void g2pc::k_blend_py_pk<4>")
    if not rec:
        return None
    return {"insts": rec["SQ_INSTS_VALU"], "cycles_per_inst": 4.0 * rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_INSTS_VALU"]}


def cpu_baseline(workload):
    """The oracle (CPU restatement of the reference, `kind: port`; pinned bit-exactly to the reference's own
    outputs by tests/test_oracle_*.py) on a bounded sample of the same workload, on this box's host cores."""
    import ref_gauss as RG
    import ref_render as RR
    from np_philox import keyed_normals
    from g2pc.synth import make_scene, make_cameras
    n, pts, ncam = 10_000, 100_000, 2
    torch.set_num_threads(min(16, max(1, os.cpu_count() or 1)))     # torch CPU ops on 3x3 batches do not scale past this
    sc = make_scene(n, 1234 + 2)
    t0 = time.perf_counter()
    cov = RG.covariances(sc.scales, sc.rots)
    nrm = RG.normals(sc.scales, sc.rots)
    xyz, colours, weights = sc.xyz, sc.colours * 255, sc.opacities
    t_render = 0.0
    if workload == "render":
        tr, intr = make_cameras(ncam)
        R = RR.PythonRendererOracle(sc.xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05)
        for name in tr:
            R(RR.get_camera(torch.tensor(tr[name]), intr[name], colour_resolution=1280))
        vis = R.get_visible_gaussians()
        xyz, cov, nrm = xyz[vis], cov[vis], nrm[vis]
        colours, weights = R.get_gaussian_colours()[vis].float(), R.max_contribution[vis]
        t_render = time.perf_counter() - t0
    cov, keep = RG.validate_covariances(cov)
    out = RG.generate_pointcloud(xyz[keep], cov[keep], colours[keep], nrm[keep], weights[keep], pts, std=2.0,
                                 exact=False, attempts=5,
                                 eps_fn=lambda gids, a, k: keyed_normals(7, gids[:, None], a, np.arange(k)[None, :]))
    dt = time.perf_counter() - t0
    what = ("%d cameras 1280x720 (%.1f s) + " % (ncam, t_render)) if workload == "render" else ""
    return {"value": out["points"].shape[0] / dt, "unit": "points/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "oracle (ref_render.py + ref_gauss.py): %d Gaussians, %scov/validate/magnitudes/distribute/sample "
                      "-> %d points, %.1f s wall; the full workload has 100x the Gaussians, %sx the cameras and 100x "
                      "the points" % (n, what, out["points"].shape[0], dt, "25" if workload == "render" else "0")}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (run it through gpurun)"
    if os.environ.get("G2PC_SHARE_GPU"):         # validation aid: several ranks on ONE GPU (gloo; RCCL refuses that)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("G2PC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    from g2pc import _native as nv
    from g2pc.synth import make_scene, make_cameras
    nv.lib()
    workload = a.workload or ("render" if have_renderer() else "sample")
    global GATHER_OUTPUT
    GATHER_OUTPUT = a.gather
    import gauss_render
    if a.t_floor is not None:
        gauss_render.DEFAULT_T_FLOOR = a.t_floor
    if a.sort_bits:
        nv.lib().g2pc_set_sort_tuning(a.sort_bits, a.sort_small)
    if a.blend_subblocks:
        gauss_render.BLEND_SUBBLOCKS = a.blend_subblocks
    if a.streams:
        gauss_render.PIPELINE_STREAMS = a.streams
    if a.no_context_pool:
        gauss_render.CONTEXT_POOL_SIZE = 0

    # ONE scene (same seed on every rank, replicated read-only); the cameras are split over the ranks (rank r renders
    # cameras r, r+N, ...), the visibility state is all-reduced, sampling is sharded by Gaussian index and the points
    # are gathered on rank 0.  Weak scaling (default): the camera rig and the point budget grow with N, so every GPU
    # keeps the configs[2] load of 50 cameras / 10M points; strong: the N = 1 job is split N ways.
    scale = world if a.scaling == "weak" else 1
    total_cameras, total_points = a.cameras * scale, a.points * scale
    scene_kw = dict(scale_lo=a.scene_scales[0], scale_hi=a.scene_scales[1]) if a.scene_scales else {}
    scene = make_scene(a.gaussians, 1234 + 3, device=device, with_sh=(workload == "render_cuda"), **scene_kw)
    cams = make_cameras(total_cameras) if workload != "sample" else None
    if cams is not None and a.camera_subset:
        keep = sorted(cams[0])[:a.camera_subset]              # profiling aid: first k of the SAME 50-camera rig
        cams = ({k: cams[0][k] for k in keep}, {k: cams[1][k] for k in keep})

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    nv.PROFILE = {}               # before the warm-up: the camera graphs are captured with their timing events
    for w in range(a.warmup):
        one_step(scene, cams, workload, total_points, device, seed=100 + w)
    nv.PROFILE.clear()
    if workload != "sample":
        gauss_render.RENDER_STATS.clear()
    sync()
    t0 = time.perf_counter()
    points = 0
    for k in range(a.steps):
        t_step = time.perf_counter()
        points += one_step(scene, cams, workload, total_points, device, seed=200 + k)
        if os.environ.get("G2PC_BENCH_DEBUG"):
            print("step %d: %.2f ms of host time" % (k, (time.perf_counter() - t_step) * 1e3), file=sys.stderr)
    sync()
    dt = time.perf_counter() - t0
    prof = nv.profile_summary()
    nv.PROFILE = None

    tot = torch.tensor([float(points), dt], dtype=torch.float64, device=device)
    if world > 1:
        import torch.distributed as dist
        pts_all = tot[0:1].clone()
        dist.all_reduce(pts_all, op=dist.ReduceOp.SUM)
        t_all = tot[1:2].clone()
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        points_all, dt_all = float(pts_all), float(t_all)
    else:
        points_all, dt_all = float(points), dt

    if rank != 0:
        return
    # dominant kernel: the single-kernel region with the largest device time (raster_front / raster_bin are
    # multi-kernel sort+scan regions whose event spans also absorb waiting under the 4-stream camera overlap;
    # profiles/*kernel_stats.csv ranks k_blend_py first)
    single = {k: v for k, v in prof.items() if k in ("raster_blend", "raster_update", "sampler_emit", "sampler_count")}
    dom = max(single.items(), key=lambda kv: kv[1][1]) if single else None
    roof = None
    if dom is not None:
        name, (launches, ms) = dom
        m_step = points / max(a.steps, 1)
        per_launch = {"sampler_emit": 56.0 * a.gaussians + 36.0 * m_step,          # read Gaussians, write the cloud
                      "sampler_count": 56.0 * a.gaussians + 4.0 * 5 * a.gaussians}.get(name)
        if name in ("raster_blend", "raster_bin", "raster_front"):
            import gauss_render
            st = gauss_render.RENDER_STATS[-launches:]
            L_avg = float(np.mean([x[0] for x in st]))
            wh = float(np.mean([x[2] for x in st]))
            passes = float(np.mean([x[1] for x in st]))
            per_launch = {"raster_blend": 56.0 * L_avg + 32.0 * wh + 40.0 * a.gaussians,     # K6 + visibility update
                          "raster_bin": (20.0 * a.gaussians + 12.0 * L_avg) + 24.0 * passes * L_avg + 8.0 * L_avg,
                          "raster_front": 88.0 * a.gaussians + 8.0 * a.gaussians + 4 * 24.0 * a.gaussians}[name]
        if per_launch is not None and ms > 0:
            ach = per_launch / (ms / launches * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(name, a), "avg_launch_ms": ms / launches,
                    "algorithmic_bytes_per_launch": per_launch,
                    "note": "k_blend_py is VALU/v_exp bound, not HBM bound (DESIGN.md §3); frac is its HBM share only"
                    if name == "raster_blend" else None}
            valu = pmc_valu(name, a)
            if valu is not None:
                # the bound that applies to this kernel: VALU issue.  wave64 instructions per launch from the committed SQ
                # PMC pass, 4 SIMDs x 256 CUs issuing one wave-instruction per 4.3 cycles (measured: ACTIVE_INST_VALU /
                # INSTS_VALU) at 2.4 GHz
                peak = 1024 * 2.4e9 / valu["cycles_per_inst"]
                roof["valu"] = {"wave_insts_per_launch": valu["insts"], "achieved": valu["insts"] / (ms / launches * 1e-3),
                                "peak": peak, "unit": "wave-instructions/s",
                                "frac": valu["insts"] / (ms / launches * 1e-3) / peak}
    stats = gauss_render.RENDER_STATS[-(len(gauss_render.RENDER_STATS) // max(a.steps, 1)):] if gauss_render.RENDER_STATS else []
    b_total = algorithmic_bytes(workload, a.gaussians, a.gaussians, points / max(a.steps, 1), cams, stats)
    job_hbm = {"algorithmic_bytes_per_step": b_total, "achieved": b_total / (dt / a.steps) / 1e9, "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": b_total / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
               "note": "rank 0's share; B_samp uses N_kept = N (upper bound)"}
    if workload == "render_cuda":
        job_hbm = None                 # per-camera instance counts are not collected on this path
    # the same job priced with the bytes of the REFERENCE's algorithm (SURVEY.md §8d worked numbers: L = 8.5 N instances
    # under 16x16 tiling, six 8-bit radix passes over 64-bit keys) -- the figure BASELINE.json's "% HBM roofline" target
    # (>= 40 %) is stated against; this design moves fewer bytes (job_hbm above)
    n_, m_ = float(a.gaussians), points / max(a.steps, 1)
    ncam = len(cams[0]) / world if cams is not None else 0            # cameras this rank rendered per step
    b_ref = 116.0 * n_ + 56.0 * n_ + 36.0 * m_ + ncam * (156.0 * n_ + (76.0 + 24.0 * 6) * 8.5 * n_ + 32.0 * 1280 * 720)
    job_hbm_ref = {"algorithmic_bytes_per_step": b_ref, "achieved": b_ref / (dt / a.steps) / 1e9, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": b_ref / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                   "note": "rank 0's share, SURVEY.md §8(d) B_total of the reference's algorithm"}
    out = {
        "metric": "coloured points/sec", "value": points_all / dt_all, "unit": "points/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_all / a.steps * 1e3, "higher_is_better": True,
        "scaling": a.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"render": "configs[2]: 1M Gaussians, 50 cameras 1280x720, 10M points, python-renderer semantics",
                                "render_cuda": "configs[4]: 1M Gaussians, 50 cameras, native-rasteriser semantics, SH degree 3, "
                                               "surface_distance_std=2.0, exact_num_points, 10M points",
                                "sample": "configs[1]: 1M Gaussians, no_render_colours, 10M points (sampling pipeline)"}[workload],
                   "gaussians": a.gaussians, "points": total_points, "points_per_gpu": total_points // world,
                   "cameras": total_cameras if workload != "sample" else 0, "cameras_per_gpu": (total_cameras // world) if workload != "sample" else 0,
                   "blend_transmittance_floor": gauss_render.DEFAULT_T_FLOOR, "parallelism": "cameras and Gaussian-index shards over %d GPU(s), RCCL all-reduce of visibility; output cloud %s" % (world, "gathered on rank 0" if a.gather else "left sharded by Gaussian index (one part per rank)")},
        "roofline": roof,
        # the whole job against the HBM roofline (B_total = B_geom + C * B_cam + B_samp, per rank)
        "job_hbm": job_hbm,
        "job_hbm_reference_algorithm": job_hbm_ref,
        "instances_per_camera": (float(np.mean([x[0] for x in gauss_render.RENDER_STATS])) if gauss_render.RENDER_STATS else None),
        "regions_ms_per_step": {k: v[1] / a.steps for k, v in sorted(prof.items())},
    }
    if not a.no_cpu_baseline and world == 1:          # the CPU baseline is a 1-GPU companion figure (rank 0, N = 1 only)
        out["cpu_baseline"] = cpu_baseline(workload)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
