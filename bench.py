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
    ap.add_argument("--workload", default=None, choices=[None, "render", "sample", "render_cuda", "config4"],
                    help="render = configs[2] (default), sample = configs[1], render_cuda = configs[4], config4 = configs[3] "
                         "(5M Gaussians, 200 cameras, 50M points: the 8-GPU job, also runnable on one GPU)")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--cameras", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also gather the sharded cloud on rank 0 inside the timed job (36 B per point over xGMI)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default) = the job the metric is quoted on (50 cameras, 10M points) split N ways; "
                         "weak = every GPU brings its own 50 cameras and 10M-point budget (50N cameras, 10M*N points)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (reference fixtures at the benchmark's scale)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads lines (configs[1] sampling job)")
    ap.add_argument("--extra-launches", type=int, default=0,
                    help="DIAGNOSTIC: empty kernels added to every camera batch's head chain (the cost of a kernel boundary)")
    ap.add_argument("--head-threads", type=int, default=0, choices=[0, 64, 128, 256],
                    help="tuning aid: block size of the head kernels without block-level cooperation (default 256)")
    ap.add_argument("--sort-bits", type=int, default=0, help="tuning aid: wide radix digit bits (8, 10 = one pass for 9-10 bit fields, or 11)")
    ap.add_argument("--sort-small", type=int, default=2 << 20, help="tuning aid: inputs up to this many keys use 4 keys/thread")
    ap.add_argument("--blend-subblocks", type=int, default=0, help="tuning aid: 8x8 sub-blocks per blend wave (1, 2, 4)")
    ap.add_argument("--scene-scales", type=float, nargs=2, default=None, metavar=("LO", "HI"),
                    help="diagnostic: Gaussian scale range of the synthetic scene (default 0.002 0.02 = SURVEY.md's)")
    ap.add_argument("--no-context-pool", action="store_true", help="tuning aid: every job captures its camera graphs anew")
    ap.add_argument("--blend-variant", type=int, default=None, choices=[None, 0, 1, 2, 3, 4, 5, 6], help="tuning aid: 2 / 3 = two-wave (unroll 4 / 2) dual-list kernel (one wave per sub-block), 1 = dual-list blend kernel, 0 = packed kernel")
    ap.add_argument("--depth-sort", default=None, choices=[None, "bucket", "radix"], help="tuning aid: depth order of the captured camera path")
    ap.add_argument("--streams", type=int, default=0, help="tuning aid: camera batches in flight (HIP streams) of the renderer")
    ap.add_argument("--cu-mask-heads", type=int, default=0, help="EXPERIMENT (libg2pc_exp.so, split pipeline modes): head streams on the first K CUs, blend streams on the other 256 - K")
    ap.add_argument("--cu-unfused", action="store_true", help="A/B aid (render_cuda): pipelined cameras through g2pc_raster_front_cu + g2pc_raster_back_cu_dev (round 4) instead of g2pc_raster_camera_cu")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the untimed region / kernel profile passes after the timed loop")
    ap.add_argument("--pipeline-mode", default=None, choices=[None, "chain", "split", "split_multi"], help="tuning aid: see gauss_render.PIPELINE_MODE")
    ap.add_argument("--slots-per-stream", type=int, default=0, help="tuning aid: batch slots per stream (gauss_render.PIPELINE_SLOTS_PER_STREAM)")
    ap.add_argument("--blend-streams", type=int, default=0, help="tuning aid (split modes): blend streams the batches alternate over")
    ap.add_argument("--camera-batch", type=int, default=0, help="tuning aid: cameras per launch sequence (1 = one camera per graph)")
    ap.add_argument("--camera-subset", type=int, default=0, help="profiling aid: render only the first k cameras of the rig")
    ap.add_argument("--walk-cap", type=int, default=0, help="DIAGNOSTIC: truncate every blend walk after this many 64-entry batches (wrong results; what do long walks cost?)")
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


ROCPROF_STATS_FILE = "profiles/r06fin_s1_kernel_stats.csv"   # rocprofv3 --kernel-trace --stats of `bench.py --streams 1 --no-parity --no-extra --no-cpu-baseline`


def rocprof_kernel_avg(region):
    """Average duration (us) of the region's kernel in the committed rocprofv3 summary, None if absent."""
    kernel = {"raster_blend": BLEND_KERNEL, "sampler_emit": "g2pc::k_emit_rows"}.get(region)
    path = os.path.join(ROOT, ROCPROF_STATS_FILE)
    if kernel is None or not os.path.isfile(path):
        return None
    import csv
    for row in csv.DictReader(open(path)):               # tools/rocprof_summary.py: kernel,calls,total_us,avg_us,...
        if kernel in row.get("kernel", ""):
            return float(row["avg_us"])
    return None


PMC_TRAFFIC_FILE = "profiles/r06fin_pmc_traffic.json"      # tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes of THIS command
PMC_SQ_FILE = "profiles/r06fin_pmc_sq.json"               # tools/pmc_kernel.py: SQ counter pass of THIS command
BLEND_KERNEL = "void g2pc::k_blend_py_dl<4>"


def _default_config(a):
    import gauss_render
    from g2pc import tiles
    sub = gauss_render.BLEND_SUBBLOCKS or tiles.SUBBLOCKS_PER_CHUNK
    return (a.gaussians, a.cameras) == (1_000_000, 50) and sub == 2 and gauss_render.DEFAULT_T_FLOOR == 1e-6


def pmc_traffic(region, a):
    """HBM bytes per launch of the region's kernel from a COMMITTED rocprofv3 PMC collection of this bench command (PMC
    counters cannot be read from inside the run; FETCH_SIZE and WRITE_SIZE need separate passes).  Returned with its
    source file so that the figure is never mistaken for a live measurement; None when the configuration differs from
    the one profiled or the profiled kernel is not the one this build launches.  Corrected as MI355X_MICROARCH.md §HBM
    prescribes for gfx950: FETCH_SIZE counts 16-byte-per-lane loads at half their bytes (doubled), WRITE_SIZE as is."""
    path = os.path.join(ROOT, PMC_TRAFFIC_FILE)
    kernel = {"raster_blend": BLEND_KERNEL, "sampler_emit": "g2pc::k_emit_rows"}.get(region)
    if kernel is None or not os.path.isfile(path) or not _default_config(a):
        return None, None
    rec = json.load(open(path)).get(kernel)
    return (rec["hbm_bytes_fetch_x2"], PMC_TRAFFIC_FILE) if rec else (None, None)


def pmc_valu(region, a):
    """VALU wave-instructions per launch of the blend kernel from the committed SQ counter pass; same conditions."""
    path = os.path.join(ROOT, PMC_SQ_FILE)
    if region != "raster_blend" or not os.path.isfile(path) or not _default_config(a):
        return None
    rec = json.load(open(path)).get(BLEND_KERNEL)
    if not rec:
        return None
    return {"insts": rec["SQ_INSTS_VALU"], "cycles_per_inst": 4.0 * rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_INSTS_VALU"],
            "wave_cycles": 4.0 * rec["SQ_WAVE_CYCLES"], "source": PMC_SQ_FILE}


GPU_CLOCK_HZ = 2.4e9
SIMDS = 1024                       # 256 CUs x 4 SIMDs
VALU_RATES_FILE = "profiles/archive/r02c_valu_rates.json"     # tools/experiments/ubench: issue rates measured on the MI355X


def valu_roof(valu, launch_s, kernel_substring):
    """The VALU side of the blend's roofline, every figure recomputable from the line: wave64 VALU instructions per launch
    (committed SQ PMC pass) over the launch duration, against
      * the ARCHITECTURAL issue rate: one wave-instruction per 2 cycles and SIMD (MI355X_MICROARCH.md, Wave scheduling: "a
        wave issues each VALU instruction over 2 cycles") = 1024 SIMDs x 2.4 GHz / 2;
      * the MIX-AWARE rate: what `v_fma_f32` sustains with 8 waves per SIMD in the issue-rate micro-benchmark
        (profiles/archive/r02c_valu_rates.json: 2.45 cycles).
    `busy_frac` is the share of the launch in which the VALU was busy at all (4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU cycles
    per instruction: round 3 printed this as `frac` -- it is not a roofline fraction).  avg_resident_waves_per_simd =
    4 x SQ_WAVE_CYCLES / 1024 SIMDs / launch cycles; vgpr / max_waves_per_simd from the shipped code object."""
    ach = valu["insts"] / launch_s
    arch = SIMDS * GPU_CLOCK_HZ / 2.0
    mix_cycles = 2.45
    try:
        rates = json.load(open(os.path.join(ROOT, VALU_RATES_FILE)))
        mix_cycles = float(rates["v_fma_f32 @8 waves/SIMD"]["cycles_at_2.4GHz"])
    except Exception:
        pass
    out = {"wave_insts_per_launch": valu["insts"], "achieved": ach, "unit": "wave-instructions/s", "source": valu["source"],
           "peak_architectural": arch, "frac_architectural": ach / arch,
           "peak_mix_aware": SIMDS * GPU_CLOCK_HZ / mix_cycles, "frac_mix_aware": ach / (SIMDS * GPU_CLOCK_HZ / mix_cycles),
           "mix_cycles_per_inst": mix_cycles, "mix_source": VALU_RATES_FILE,
           "busy_frac": ach / (SIMDS * GPU_CLOCK_HZ / valu["cycles_per_inst"]),
           "avg_resident_waves_per_simd": valu["wave_cycles"] / SIMDS / (launch_s * GPU_CLOCK_HZ)}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_meta import kernel_meta
        km = kernel_meta(kernel_substring)
        if km:
            out.update(vgpr=km["vgpr_count"], lds_bytes=km["group_segment_fixed_size"], max_waves_per_simd=km["max_waves_per_simd"])
    except Exception:
        pass
    return out


def cpu_baseline(workload, full_n, full_cams, full_points, n=200_000, pts=1_000_000):
    """CPU baseline on THIS box's host cores, rank 0, N = 1, a bounded sample (~10-30 s) of the same workload.

    kind "port": the untouched reference is Python and cannot travel to the GPU box (no /root/reference there), so
    what is timed is oracle/ref_render.py + ref_gauss.py -- a restatement of the reference's python renderer and
    sampler that is pinned BIT-EXACTLY to the reference's own outputs (tests/test_oracle_*.py) and issues the same
    torch CPU ops.  Sample: the renderer on ONE 1280x720 camera at N = 200 k Gaussians (a fifth of the workload's) and
    the sampler on 1 M points; the figure is the workload's point count over the extrapolated job time
    C * t_camera * (N_full / N_sample) + t_sampler * (M_full / M_sample) -- linear in N is generous to the CPU
    (SURVEY.md §6 measures 2 s -> 12.5 s -> ~2 min per camera for 10 k -> 100 k -> 1 M).  The untouched reference itself,
    timed in the authoring container on the full-size scene (8 threads), is quoted from the parity fixture."""
    import ref_gauss as RG
    import ref_render as RR
    from np_philox import keyed_normals
    from g2pc.synth import make_scene, make_cameras
    # torch's CPU kernels on batches of 3x3 / per-tile tensors stop scaling (and with hundreds of threads collapse) beyond
    # a few tens of threads: cap at 32
    threads = max(1, min(32, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    if workload != "sample" and n > 20_000:
        # bound the wall time whatever the host: a 20 k-Gaussian camera first, the full sample only if that projects to < 25 s
        sc0 = make_scene(20_000, 1234 + 3)
        tr0, intr0 = make_cameras(50)
        R0 = RR.PythonRendererOracle(sc0.xyz, sc0.opacities.unsqueeze(1), sc0.colours.double(), RG.covariances(sc0.scales, sc0.rots), threshold=0.05)
        tp = time.perf_counter()
        R0(RR.get_camera(torch.tensor(tr0[sorted(tr0)[17]]), intr0[sorted(tr0)[17]], colour_resolution=1280))
        probe = time.perf_counter() - tp
        while n > 20_000 and probe * (n / 20_000) > 25.0:
            n //= 2
    sc = make_scene(n, 1234 + 3)
    t0 = time.perf_counter()
    cov = RG.covariances(sc.scales, sc.rots)
    nrm = RG.normals(sc.scales, sc.rots)
    t_geom = time.perf_counter() - t0
    xyz, colours, weights = sc.xyz, sc.colours * 255, sc.opacities
    t_cam = 0.0
    if workload != "sample":
        tr, intr = make_cameras(50)
        name = sorted(tr)[17]
        R = RR.PythonRendererOracle(sc.xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05)
        t1 = time.perf_counter()
        R(RR.get_camera(torch.tensor(tr[name]), intr[name], colour_resolution=1280))
        t_cam = time.perf_counter() - t1
        vis = R.get_visible_gaussians()
        xyz, cov, nrm = xyz[vis], cov[vis], nrm[vis]
        colours, weights = R.get_gaussian_colours()[vis].float(), R.max_contribution[vis]
    t2 = time.perf_counter()
    cov, keep = RG.validate_covariances(cov)
    out = RG.generate_pointcloud(xyz[keep], cov[keep], colours[keep], nrm[keep], weights[keep], pts, std=2.0,
                                 exact=False, attempts=5,
                                 eps_fn=lambda gids, a, k: keyed_normals(7, gids[:, None], a, np.arange(k)[None, :]))
    t_samp = time.perf_counter() - t2
    m = int(out["points"].shape[0])
    est = full_cams * t_cam * (full_n / n) + (t_geom + t_samp) * (full_points / max(m, 1))
    res = {"value": full_points / est, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "oracle port of the reference (ref_render.py + ref_gauss.py, bit-pinned to it): N = %d Gaussians, "
                     "%s sampler %d points in %.1f s (geometry %.2f s); value = %d points / (%d cameras x t_camera x %.0f "
                     "+ t_sampler x %.0f) = extrapolated job time %.0f s"
                     % (n, ("1 camera 1280x720 in %.1f s," % t_cam) if workload != "sample" else "no cameras,", m, t_samp,
                        t_geom, full_points, full_cams, full_n / n, full_points / max(m, 1), est),
           "measured_seconds": {"camera": t_cam, "sampler": t_samp, "geometry": t_geom}}
    fx = os.path.join(ROOT, "tests", "golden", "render_py_cfg2_1m.npz")
    fs = os.path.join(ROOT, "tests", "golden", "sample_cfg2_1m.npz")
    if os.path.isfile(fx) and os.path.isfile(fs):
        g, gs = np.load(fx), np.load(fs)
        spc = float(np.min(g["seconds_per_camera"]))
        res["untouched_reference_authoring_container"] = {
            "seconds_per_camera_1M_gaussians_1280x720": spc, "sampler_seconds_10M_points": float(gs["sample_seconds"]),
            "threads": int(g["threads"]),
            "points_per_s_configs2_extrapolated": full_points / (full_cams * spc + float(gs["sample_seconds"])) if workload != "sample"
            else full_points / float(gs["sample_seconds"]),
            "note": "the reference itself (renderer_type=python under oracle/ref_shim.py) on the full-size scene, timed by "
                    "oracle/make_golden.py in the authoring container -- a different host than this GPU box"}
    return res


def extra_sample_line(a, device):
    """BASELINE configs[1] (sampling pipeline only) timed the same way as the main workload, as an extra line."""
    from g2pc.synth import make_scene
    scene = make_scene(a.gaussians, 1234 + 2, device=device)
    for w in range(2):
        one_step(scene, None, "sample", a.points, device, seed=300 + w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts, k = 0, 5
    for i in range(k):
        pts += one_step(scene, None, "sample", a.points, device, seed=400 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    b = algorithmic_bytes("sample", a.gaussians, a.gaussians, pts / k, None, [])
    return {"config": "configs[1]: %d Gaussians, no_render_colours, %d points (cov build -> validate -> magnitudes -> "
                      "distribute -> sample)" % (a.gaussians, a.points),
            "value": pts / k / dt, "unit": "points/s", "ms_per_step": dt * 1e3, "steps": k,
            "job_hbm": {"algorithmic_bytes_per_step": b, "achieved": b / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": b / dt / 1e9 / HBM_PEAK_GBS}}


def relaunch_for_gpus(a):
    """`python bench.py --gpus N` without a torch.distributed environment: re-execute under torch.distributed.run (one rank
    per GPU, RCCL).  Never falls through to a one-rank run that would print n_gpus: 1 for an N-GPU request."""
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus > 1 and a.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)


def extra_render_line(a, device, workload, t_floor=None, steps=3, warm=2, sizes=None):
    """BASELINE configs[4] (`render_cuda`: native-rasteriser semantics, SH degree 3, surface distance, exact points) or the
    to-the-letter mode of configs[2] (`render`, t_floor = 0: no transmittance floor, k_blend_py_pk), timed like the main
    workload -- `steps` whole jobs between synchronisations, inputs resident -- with the roofline of ITS dominant kernel:
    launch duration from HIP events on the launch stream around that kernel alone (one camera in flight), algorithmic bytes
    56 L + 32 W H per launch."""
    import gauss_render
    from g2pc import _native as nv
    from g2pc.synth import make_scene, make_cameras
    saved_floor = gauss_render.DEFAULT_T_FLOOR
    if t_floor is not None:
        gauss_render.DEFAULT_T_FLOOR = t_floor
    if sizes is not None:                      # (gaussians, cameras, points) other than the main workload's
        import argparse
        a = argparse.Namespace(**dict(vars(a), gaussians=sizes[0], cameras=sizes[1], points=sizes[2]))
    gauss_render.clear_context_pool()
    try:
        scene = make_scene(a.gaussians, 1234 + 3, device=device, with_sh=(workload == "render_cuda"))
        cams = make_cameras(a.cameras)
        for w in range(warm):
            one_step(scene, cams, workload, a.points, device, seed=500 + w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts = 0
        for i in range(steps):
            pts += one_step(scene, cams, workload, a.points, device, seed=600 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out = {"value": pts / steps / dt, "unit": "points/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": warm}
        if sizes is not None:
            # how the cameras of one (untimed) job went through the renderer: children of split leaves rendered, cameras whose
            # quad-tree levels the host had to walk, cameras rendered again for want of room in their graph
            st = per_rank_stages(scene, cams, workload, a.points, device, 1)[0]
            out["quad_tree"] = {k: st.get(k) for k in ("cameras", "split_children", "child_pass_cameras", "host_driven_cameras",
                                                       "rerendered_cameras")}
            out["stages_ms"] = {k: st.get(k) for k in ("setup_ms", "camera_loop_ms", "fixed_ms", "sample_ms")}
        # the dominant kernel alone on the device: three cameras of the rig, one at a time
        names = sorted(cams[0])[:3]
        wh = 1280 * 720
        if workload == "render_cuda":
            import camera_handler
            from gauss_handler import Gaussians
            g = Gaussians(scene.xyz, scene.scales, scene.rots, scene.colours.clone(), scene.opacities, shs=scene.shs)
            R = gauss_render.get_renderer("cuda", g.xyz, g.opacities.unsqueeze(1), g.colours, g.covariances, shs=g.shs,
                                          visible_gaussian_threshold=0.05, surface_distance_std=2.0, calculate_surface_distance=True)
            nv.PROFILE = {}
            Ls = []
            for rep in range(2):
                nv.PROFILE.clear()
                Ls = []
                for i, nm in enumerate(names):
                    cam = camera_handler.get_camera("cuda", torch.tensor(cams[0][nm]), cams[1][nm], colour_resolution=1280, sh_degree=3)
                    c, campos, mask = R._camera(cam)
                    sc = R._sync
                    R._front(sc, c, campos, cam.sh_degree)
                    L = int(sc.offsets[R.n].item())
                    Ls.append(L)
                    R._back(sc, c, mask, L, i, 1, "raster_bin_cu")
                    R._back(sc, c, mask, L, i, 2, "raster_blend_cu")
                    R._back(sc, c, mask, L, i, 4, "raster_update_cu")
                torch.cuda.synchronize()
            prof = nv.profile_summary()
            nv.PROFILE = None
            kernel, region, L_avg = "g2pc::k_blend_cu", "raster_blend_cu", float(np.mean(Ls))
            del R
        else:
            saved = gauss_render.PIPELINE_STREAMS
            gauss_render.PIPELINE_STREAMS = 1
            gauss_render.RENDER_STATS.clear()
            sub = ({k: cams[0][k] for k in names}, {k: cams[1][k] for k in names})
            nv.PROFILE = {}
            one_step(scene, sub, workload, a.points, device, seed=700)
            torch.cuda.synchronize()
            nv.PROFILE.clear()
            gauss_render.RENDER_STATS.clear()
            one_step(scene, sub, workload, a.points, device, seed=701)
            torch.cuda.synchronize()
            prof = nv.profile_summary()
            nv.PROFILE = None
            gauss_render.PIPELINE_STREAMS = saved
            kernel, region = "g2pc::k_blend_py_pk<4>" if t_floor == 0.0 else BLEND_KERNEL, "raster_blend"
            L_avg = float(np.mean([x[0] for x in gauss_render.RENDER_STATS])) if gauss_render.RENDER_STATS else float("nan")
            gauss_render.RENDER_STATS.clear()
        if region in prof and prof[region][0] > 0:
            launches, ms = prof[region]
            per_launch = 56.0 * L_avg + 32.0 * wh
            ach = per_launch / (ms / launches * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms / launches, "launches_timed": launches,
                               "algorithmic_bytes_per_launch": per_launch, "instances_per_camera": L_avg,
                               "avg_launch_ms_source": "HIP events on the launch stream around the kernel alone, cameras %s of the rig" % names}
        return out
    finally:
        gauss_render.DEFAULT_T_FLOOR = saved_floor
        gauss_render.clear_context_pool()


def per_rank_stages(scene, cams, workload, total_points, device, world):
    """One UNTIMED job with the device synchronised at every stage boundary (convert_gaussians_to_pc(stage_times=...)),
    gathered from all ranks: what each rank spent in its camera loop, the visibility exchange, the per-job fixed work and
    its sampling shard -- the breakdown a measured 1 -> N curve is read against."""
    from gauss_handler import Gaussians
    from gauss_to_pc import convert_gaussians_to_pc
    g = Gaussians(scene.xyz, scene.scales, scene.rots, scene.colours.clone(), scene.opacities, shs=scene.shs)
    st = {}
    transforms, intr = cams if cams is not None else (None, None)
    convert_gaussians_to_pc(g, transforms, intr, None, settings(workload, total_points, device), seed=950,
                            render_shs=(workload == "render_cuda"), stage_times=st)
    st = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}
    if world > 1:
        import torch.distributed as dist
        st["rank"] = dist.get_rank()
        allst = [None] * world
        dist.all_gather_object(allst, st)
        return allst
    st["rank"] = 0
    return [st]


def main():
    a = parse()
    relaunch_for_gpus(a)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    emulate = bool(os.environ.get("G2PC_BENCH_EMULATE")) and not torch.cuda.is_available()
    assert torch.cuda.is_available() or emulate, "bench.py needs an MI355X (run it through gpurun)"
    if emulate:
        # authoring-container dry run of THIS script's code paths (argument handling, JSON assembly, parity / baseline
        # plumbing) on the CPU emulator of the kernels at toy sizes; the numbers it prints mean nothing
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_util import build_emu
        from g2pc import _native as _nv
        _nv._inject_for_tests(build_emu())
        torch.cuda.synchronize = lambda *a, **k: None
        device = torch.device("cpu")
    else:
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
    import gauss_render  # noqa: F401  (registers the rasteriser prototypes)
    if os.environ.get("G2PC_DUMMY_STREAMS") and not emulate:
        # experiment: does the position of the renderer's streams in torch's stream pool (-> hardware queue mapping) explain
        # the slow first context?
        _dummy_streams = [torch.cuda.Stream(device) for _ in range(int(os.environ["G2PC_DUMMY_STREAMS"]))]
    if os.environ.get("G2PC_PREALLOC_GB") and not emulate:
        # experiment: let torch's caching allocator obtain ONE large device block first, so that everything the first job
        # allocates is carved out of it instead of coming from many separate hipMalloc calls
        big = torch.empty((int(float(os.environ["G2PC_PREALLOC_GB"]) * (1 << 30)),), dtype=torch.uint8, device=device)
        del big
    workload = a.workload or ("render" if have_renderer() else "sample")
    config4 = workload == "config4"
    if config4:                                   # configs[3]: same pipeline as "render", BASELINE's 8-GPU sizes
        workload = "render"
        if (a.gaussians, a.cameras, a.points) == (1_000_000, 50, 10_000_000):
            a.gaussians, a.cameras, a.points = 5_000_000, 200, 50_000_000
    global GATHER_OUTPUT
    GATHER_OUTPUT = a.gather
    import gauss_render
    if a.t_floor is not None:
        gauss_render.DEFAULT_T_FLOOR = a.t_floor
    # tuning / diagnostic knobs: entry points of -DG2PC_EXPERIMENTS builds only (run through tools/experiments/ab_lib.py with
    # 3dgs-to-pc_amd/g2pc/libg2pc_exp.so); the product library has no process-global state and nv.experiments() raises for it
    if a.sort_bits or a.sort_small != (2 << 20):
        nv.experiments().g2pc_set_sort_tuning(a.sort_bits or 8, a.sort_small)
    if a.head_threads:
        nv.experiments().g2pc_debug_set_head_threads(a.head_threads)
    if a.extra_launches:
        nv.experiments().g2pc_debug_set_extra_launches(a.extra_launches)
    if a.blend_variant is not None:
        nv.experiments().g2pc_set_blend_variant(a.blend_variant)
    if a.walk_cap:
        nv.experiments().g2pc_debug_set_walk_cap(a.walk_cap)
    if a.depth_sort:
        nv.experiments().g2pc_set_depth_sort(1 if a.depth_sort == "bucket" else 0)
    if a.blend_subblocks and a.blend_subblocks != 2:
        nv.experiments()                           # (1 or 4 sub-blocks per wave: the scalar blend of experiments builds)
    if a.blend_subblocks:
        gauss_render.BLEND_SUBBLOCKS = a.blend_subblocks
    if a.streams:
        gauss_render.PIPELINE_STREAMS = a.streams
        import gaussian_pointcloud_rasterization as _gpr
        _gpr.PIPELINE_STREAMS = a.streams
    if a.cu_mask_heads:
        import ctypes as _C
        import torch as _t
        K, _L = int(a.cu_mask_heads), nv.experiments()

        def _masked(device, kind, K=K):
            bits = [(i < K) if kind == "head" else (i >= K) for i in range(256)]
            words = (_C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
            st = _C.c_void_p(None)
            nv.check(_L.g2pc_debug_stream_create_cu_mask(words, 8, _C.byref(st)), "stream_create_cu_mask")
            return _t.cuda.ExternalStream(st.value, device=device)
        gauss_render.STREAM_FACTORY = _masked
    if a.cu_unfused:
        import gaussian_pointcloud_rasterization as _gpr2
        _gpr2.FUSED_CAMERA_CALL = False
    if a.camera_batch:
        gauss_render.CAMERA_BATCH = a.camera_batch
    if a.pipeline_mode:
        gauss_render.PIPELINE_MODE = a.pipeline_mode
    if a.blend_streams:
        gauss_render.PIPELINE_BLEND_STREAMS = a.blend_streams
    if a.slots_per_stream:
        gauss_render.PIPELINE_SLOTS_PER_STREAM = a.slots_per_stream
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
    # (the emulator dry run renders miniature images: a 1280 x 720 camera is 921 600 emulated lanes per blend)
    cam_kw = dict(width=128, height=72, focal=110.0) if emulate else {}
    cams = make_cameras(total_cameras, **cam_kw) if workload != "sample" else None
    if cams is not None and a.camera_subset:
        keep = sorted(cams[0])[:a.camera_subset]              # profiling aid: first k of the SAME 50-camera rig
        cams = ({k: cams[0][k] for k in keep}, {k: cams[1][k] for k in keep})

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Process warm-up (g2pc/warmup.py): a miniature of the pipeline loads every code object the job touches -- part of
    # "library load", which the metric excludes; the CLI runs it while the host parses its input files.
    warmup_s = 0.0
    if not emulate and not os.environ.get("G2PC_NO_WARMUP"):
        from g2pc.warmup import warmup
        warmup_s = warmup(device, ("cuda",) if workload == "render_cuda" else ("python",))
    # The timed loop runs the PRODUCTION path: no region events, every camera one unsplit hipGraph.  Region times and the
    # dominant kernel's launch duration are collected afterwards, in separate untimed passes (profile_pass below).
    nv.PROFILE = None
    first_job_ms, first_job_points = None, 0
    for w in range(a.warmup):
        t_w = time.perf_counter()
        n_w = one_step(scene, cams, workload, total_points, device, seed=100 + w)
        if w == 0:
            torch.cuda.synchronize()
            first_job_ms = (time.perf_counter() - t_w) * 1e3      # what a one-shot `python gauss_to_pc.py ...` pays (graph capture,
            first_job_points = n_w                                # first-use allocations, no pooled context), library load excluded
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
    timed_stats = list(gauss_render.RENDER_STATS)

    def profile_pass(streams):
        """One untimed job with HIP events around every region (on the stream the kernels are launched on; with events on,
        a camera graph stops before the blend and the blend is issued directly, so that it can be bracketed alone).
        streams = 1: one camera at a time -- the event span of a single-kernel region IS that kernel's duration."""
        saved = (gauss_render.PIPELINE_STREAMS,)
        if streams:
            gauss_render.PIPELINE_STREAMS = streams
        nv.PROFILE = {}
        one_step(scene, cams, workload, total_points, device, seed=900)          # (re)captures the split graphs
        torch.cuda.synchronize()
        nv.PROFILE.clear()
        one_step(scene, cams, workload, total_points, device, seed=901)
        torch.cuda.synchronize()
        res = nv.profile_summary()
        nv.PROFILE = None
        gauss_render.PIPELINE_STREAMS = saved[0]
        return res

    prof, prof_alone = {}, {}
    if world == 1 and not emulate and not a.no_profile_pass:
        prof = profile_pass(0)                           # production stream count: spans overlap across streams
        prof_alone = profile_pass(1) if workload == "render" else prof
        gauss_render.RENDER_STATS[:] = timed_stats

    per_rank = None
    if not a.no_profile_pass and workload != "sample":
        per_rank = per_rank_stages(scene, cams, workload, total_points, device, world)   # every rank takes part (collectives inside)
        gauss_render.RENDER_STATS[:] = timed_stats

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
    single = {k: v for k, v in prof_alone.items() if k in ("raster_blend", "raster_update", "sampler_emit", "sampler_count")}
    dom = max(single.items(), key=lambda kv: kv[1][1]) if single else None
    roof = None
    if dom is not None:
        name, (launches, ms) = dom
        m_step = points / max(a.steps, 1)
        per_launch = {"sampler_emit": 56.0 * a.gaussians + 36.0 * m_step,          # read Gaussians, write the cloud
                      "sampler_count": 56.0 * a.gaussians + 4.0 * 5 * a.gaussians}.get(name)
        if name in ("raster_blend", "raster_bin", "raster_front"):
            import gauss_render
            st = gauss_render.RENDER_STATS[-(len(gauss_render.RENDER_STATS) // max(a.steps, 1)):]
            L_avg = float(np.mean([x[0] for x in st]))
            wh = float(np.mean([x[2] for x in st]))
            passes = float(np.mean([x[1] for x in st]))
            per_launch = {"raster_blend": 56.0 * L_avg + 32.0 * wh,     # 44 L read + 12 L visibility RMW + 32 W H
                          "raster_bin": (20.0 * a.gaussians + 12.0 * L_avg) + 24.0 * passes * L_avg + 8.0 * L_avg,
                          "raster_front": 88.0 * a.gaussians + 8.0 * a.gaussians + 4 * 24.0 * a.gaussians}[name]
        if per_launch is not None and ms > 0:
            ach = per_launch / (ms / launches * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(name, a)[0], "traffic_source": pmc_traffic(name, a)[1],
                    "avg_launch_ms": ms / launches,
                    "avg_launch_ms_source": "HIP events on the launch stream, separate untimed pass with ONE camera in flight "
                                            "(the kernel alone on the device); under the production %d-stream overlap the same "
                                            "region spans %.4f ms" % (gauss_render.PIPELINE_STREAMS,
                                                                     prof[name][1] / max(prof[name][0], 1) if name in prof else float("nan")),
                    "kernel_avg_us_rocprof": rocprof_kernel_avg(name), "kernel_avg_us_rocprof_source": ROCPROF_STATS_FILE,
                    "algorithmic_bytes_per_launch": per_launch,
                    "note": "the blend is bound by VALU issue at low wave residency, not by HBM (DESIGN.md §4); frac is its HBM share only"
                    if name == "raster_blend" else None}
            valu = pmc_valu(name, a)
            if valu is not None:
                roof["valu"] = valu_roof(valu, ms / launches * 1e-3, BLEND_KERNEL.split("::")[-1])     # the timed INSTANCE, <4>
    stats = gauss_render.RENDER_STATS[-(len(gauss_render.RENDER_STATS) // max(a.steps, 1)):] if gauss_render.RENDER_STATS else []
    b_total = algorithmic_bytes(workload, a.gaussians, a.gaussians, points / max(a.steps, 1), cams, stats)
    job_hbm = {"algorithmic_bytes_per_step": b_total, "achieved": b_total / (dt / a.steps) / 1e9, "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": b_total / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
               "note": "rank 0's share; B_samp uses N_kept = N (upper bound)"}
    if workload == "render_cuda":
        job_hbm = None                 # per-camera instance counts are not collected on this path
    out = {
        "metric": "coloured points/sec", "value": points_all / dt_all, "unit": "points/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_all / a.steps * 1e3, "higher_is_better": True,
        "scaling": a.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[3]: 5M Gaussians, 200 cameras 1280x720, 50M points, python-renderer semantics" if config4 else
                               {"render": "configs[2]: 1M Gaussians, 50 cameras 1280x720, 10M points, python-renderer semantics",
                                "render_cuda": "configs[4]: 1M Gaussians, 50 cameras, native-rasteriser semantics, SH degree 3, "
                                               "surface_distance_std=2.0, exact_num_points, 10M points",
                                "sample": "configs[1]: 1M Gaussians, no_render_colours, 10M points (sampling pipeline)"}[workload],
                   "gaussians": a.gaussians, "points": total_points, "points_per_gpu": total_points // world,
                   "cameras": total_cameras if workload != "sample" else 0, "cameras_per_gpu": (total_cameras // world) if workload != "sample" else 0,
                   "blend_transmittance_floor": gauss_render.DEFAULT_T_FLOOR, "parallelism": "cameras and Gaussian-index shards over %d GPU(s), RCCL all-reduce of visibility; output cloud %s" % (world, "gathered on rank 0" if a.gather else "left sharded by Gaussian index (one part per rank)")},
        "roofline": roof,
        # the whole job against the HBM roofline (B_total = B_geom + C * B_cam + B_samp, per rank)
        "job_hbm": job_hbm,
        "instances_per_camera": (float(np.mean([x[0] for x in gauss_render.RENDER_STATS])) if gauss_render.RENDER_STATS else None),
        "regions_ms_per_step": {k: v[1] for k, v in sorted(prof.items())},
        "regions_note": "HIP-event spans of ONE untimed job after the timed loop (production stream count: spans of different "
                        "cameras overlap, their sum may exceed ms_per_step); the timed loop itself records no events",
        "per_rank": per_rank,
        "per_rank_note": "one untimed job with the device synchronised at every stage boundary: milliseconds per rank for renderer "
                         "set-up, the camera loop (cameras = how many this rank rendered), the RCCL visibility exchange, the per-job "
                         "fixed work (getters, cull, filter, validate) and the rank's sampling shard",
        "rccl_world_size": (__import__("torch").distributed.get_world_size() if world > 1 else 1),
        "first_job_ms": first_job_ms,
        "process_warmup_ms": warmup_s * 1e3,
        "first_job_points_per_s": (first_job_points * world / (first_job_ms * 1e-3)) if first_job_ms else None,
    }
    if world == 1 and workload == "render" and not config4 and not a.no_extra and not a.camera_subset:
        out["extra_workloads"] = {"sample": extra_sample_line(a, device)}
        if not emulate and (a.gaussians, a.cameras) == (1_000_000, 50):
            # the figures README / DESIGN quote beside the headline, measured by THIS command on THIS box
            x = extra_render_line(a, device, "render_cuda")
            x["config"] = ("configs[4]: 1M Gaussians, 50 cameras, native-rasteriser semantics, SH degree 3, surface_distance_std=2.0, "
                           "exact_num_points, 10M points")
            out["extra_workloads"]["render_cuda"] = x
            x = extra_render_line(a, device, "render", t_floor=0.0)
            x["config"] = ("configs[2] with blend_transmittance_floor = 0: the reference's python-renderer semantics to the letter "
                           "(no visit dropped, k_blend_py_pk)")
            out["extra_workloads"]["exact"] = x
            try:
                x = extra_render_line(a, device, "render", steps=2, warm=1, sizes=(5_000_000, 200, 50_000_000))
            except Exception as e:                 # (the largest job of the line: never at the price of the line itself)
                x = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                torch.cuda.empty_cache()
            x["config"] = ("configs[3] on ONE GPU: 5M Gaussians, 200 cameras 1280x720, 50M points, python-renderer semantics (the "
                           "8-GPU job of BASELINE.json; more than half of its cameras overload a leaf: on-demand child pass)")
            out["extra_workloads"]["config4"] = x
    if world == 1 and workload == "render" and not a.no_parity:
        # parity gates, outside the timed region: the same scene, cameras 0 and 17 of the same rig, against
        # outputs of the untouched reference (tests/golden/*_cfg2_1m.npz); tools/parity_cfg2.py documents every key
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import parity_cfg2
        tag = "mini" if emulate else "1m"
        if parity_cfg2.available(tag) and (a.gaussians == 1_000_000 or emulate):
            gauss_render.clear_context_pool()
            full = parity_cfg2.run(str(device), tag=tag)
            out["parity"] = {k: full.get(k) for k in (
                "mask_flips", "near_threshold_1e-5", "contrib_max", "contrib_frac_gt_1e-4", "colour_max", "colour_frac_gt_1e-4",
                "colour_off_gaussians", "colour_compared_gaussians",
                "image_max", "image_frac_gt_1e-4", "culled_equal", "ppg_mismatch_given_ref_contrib", "ppg_mismatch_end_to_end",
                "ppg_max_abs_diff_given_ref_contrib", "ppg_max_abs_diff_end_to_end",
                "sample_points", "sample_points_ref", "sample_xyz_max", "sample_rgb_max", "sample_rows_compared", "sample_rows_unmatched", "sample_rows_order_shifted", "cameras",
                "k1", "cov3d_rows_differing", "camera_matrix_bits_differing", "reference_tie_rule", "reference_tie_spread",
                "gaussians", "resolution", "t_floor", "oracle", "check_seconds")}
            # two statements, never one: point quotas from the REFERENCE's contributions (isolates magnitudes + distribute_points)
            # and from our own render end to end (float64 closed-form eigenvalues here, float32 LAPACK there: a few quotas +-1)
            out["parity"]["ppg_equal_given_ref_contrib"] = full.get("ppg_mismatch_given_ref_contrib") == 0
            out["parity"]["ppg_equal_end_to_end"] = full.get("ppg_mismatch_end_to_end") == 0
            # every end-to-end quota difference is a rounding-boundary case or it is not (tools/parity_cfg2.py::explain_quota_flips)
            out["parity"]["ppg_flips_explained"] = full.get("ppg_flips_explained_end_to_end")
            out["parity"]["ppg_flip_margin_over_bound"] = full.get("ppg_flip_margin_over_bound_end_to_end")
        # ... and THE BENCHMARKED JOB ITSELF: all 50 cameras through the production path (convert_gaussians_to_pc: pipelined
        # cameras, graph replays, deferred colour resolve -- what the timed loop above ran) against the untouched reference over
        # the same 50 cameras (tests/golden/render_py_cfg2_1m_all50.npz; tools/parity_all50.py documents every key)
        import parity_all50
        tag50 = "mini_all6" if emulate else "1m_all50"
        if "parity" in out and parity_all50.available(tag50) and (a.gaussians == 1_000_000 or emulate):
            gauss_render.clear_context_pool()
            r50 = parity_all50.run(str(device), tag=tag50)
            gauss_render.clear_context_pool()
            out["parity"]["all50"] = {k: r50.get(k) for k in (
                "cameras", "path", "pipeline", "t_floor", "mask_flips", "near_threshold_1e-5", "visible", "contrib_max",
                "contrib_frac_gt_1e-4", "contrib_compared", "winner_camera_mismatch", "winner_camera_compared", "colour_max",
                "colour_max_same_winner", "colour_off_gaussians", "colour_compared_gaussians", "culled_equal", "keep_equal", "kept",
                "ppg_mismatch_given_ref_contrib", "ppg_mismatch_end_to_end", "ppg_flips_explained", "ppg_max_abs_diff_end_to_end",
                "ppg_flip_margin_over_bound_max", "sample_points", "sample_points_ref", "sample_rows_compared",
                "sample_rows_unmatched", "sample_xyz_max", "sample_rgb_max", "sample_rgb_rows_gt_1e-4", "sample_rows_order_shifted",
                "ppg_flips_explained_given_ref_contrib", "cov3d_rows_differing", "camera_matrix_bits_differing", "reference_cpu_seconds_per_camera_mean",
                "reference_cpu_threads", "oracle", "check_seconds")}
            out["parity"]["cameras_all50"] = r50.get("cameras")
        # ... and the reference's DATA-DEPENDENT quad-tree (leaves over max_gaussians_per_tile split, gauss_render.py:319-335)
        # against outputs of the untouched reference on a scene that meets it: 150 000 Gaussians crowded into 160 x 96 pixels,
        # the same fixture tests/test_gpu_quadtree.py holds the renderer to (tests/render_checks.py::run_split_fixture)
        fx = os.path.join(ROOT, "tests", "golden", "render_py_split_60k.npz")
        if "parity" in out and os.path.isfile(fx) and not emulate:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from render_checks import run_split_fixture
            gauss_render.clear_context_pool()
            q = run_split_fixture(os.path.dirname(fx), "60k", device=str(device))
            gauss_render.clear_context_pool()
            out["parity"]["quadtree_split_fixture"] = {
                "fixture": "tests/golden/render_py_split_60k.npz", "children_of_split_leaves": q["split_leaves"],
                "contrib_max": q["contribution"], "image_max": q["image"], "colour_max_above_1e-30": q["colour"],
                "mask_flips": q["flips"], "colours_differing_below_1e-30": q["colour_off_tiny"]}
    if not a.no_cpu_baseline and world == 1:          # the CPU baseline is a 1-GPU companion figure (rank 0, N = 1 only)
        out["cpu_baseline"] = cpu_baseline(workload, a.gaussians, a.cameras if workload != "sample" else 0, a.points,
                                           **(dict(n=3000, pts=20_000) if emulate else {}))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
