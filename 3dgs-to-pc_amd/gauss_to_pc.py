"""
Drop-in mirror of the reference's ``gauss_to_pc.py``: same CLI flags, ``GaussPointCloudSettings``,
``PointCloudData`` and function names (``distribute_points``, ``mahalanobis``, ``calculate_bin_sizes``,
``sample_from_multivariate_normal``, ``create_new_gaussian_points``, ``generate_pointcloud``,
``convert_3dgs_to_pc``, ``config_parser``, ``main``), with the arithmetic in libg2pc.so (HIP, gfx950).
Reference lines are cited as ``gauss_to_pc.py:NN`` (= /root/reference/gauss_to_pc.py).

Deliberate, documented differences (DESIGN.md "Sampler"):
  * RNG: the reference draws from torch's global, never-seeded Philox stream; here every draw is
    eps(seed, gaussian id, attempt, k) from a keyed Philox4x32-10 (``SAMPLER_SEED``, or ``seed=`` kwargs),
    so runs are reproducible and independent of sharding.
  * all bins are processed by one set of kernel launches; the output ORDER is the reference's.
  * colours / normals of the cloud are float32 (the reference's float64 comes from its empty float64
    accumulators, gauss_to_pc.py:316-318); set ``REFERENCE_DTYPES = True`` to get float64 back.
  * device-agnostic: ``device`` arguments are honoured, nothing hard-codes "cuda:0".
"""
import argparse
import gc
import os
import sys
from math import floor
from typing import NamedTuple

import numpy as np
import torch

from g2pc import ops
from gauss_handler import Gaussians
from gauss_render import get_renderer
from camera_handler import get_camera

COLOR_QUALITY_OPTIONS = {"tiny": 180, "low": 360, "medium": 720, "high": 1280, "ultra": 1920, "original": None}

SAMPLER_SEED = 0
CAMERA_EPOCH = 255          # at most this many cameras per epoch of the renderer's camera-order field (multi-GPU camera sharding)
REFERENCE_DTYPES = False


class GaussPointCloudSettings(NamedTuple):
    """gauss_to_pc.py:26-60 (field order and meaning unchanged)."""
    renderer_type: str
    num_points: int
    prioritise_visible_gaussians: bool
    mahalanobis_distance_std: float
    camera_skip_rate: int
    render_colours: bool
    min_opacity: float
    bounding_box_min: list
    bounding_box_max: list
    calculate_normals: bool
    cull_large_percentage: float
    remove_unrendered_gaussians: bool
    colour_resolution: int
    max_sh_degree: int
    exact_num_points: int
    visibility_threshold: float
    surface_distance_std: float
    generate_mesh: bool
    quiet: bool
    device: str


class PointCloudData(NamedTuple):
    points: torch.Tensor
    colours: torch.Tensor
    normals: torch.Tensor


def distribute_points(gaussian_sizes, num_points):
    """gauss_to_pc.py:73-90 -- float64 points per Gaussian, incl. the zero-fill (and its negative-slice quirk)."""
    return ops.distribute_points(gaussian_sizes, num_points)[0]


def mahalanobis(means, samples, covs):
    """gauss_to_pc.py:92-103."""
    return ops.mahalanobis(means, samples, covs)


def calculate_bin_sizes(points_per_gaussian):
    """gauss_to_pc.py:105-138 -- histogram on the device, the 100-odd-element heuristic on the host."""
    ppg = points_per_gaussian.to(torch.int32).contiguous()
    length = int(ppg.max().item()) + 1
    hist = ops.bincount(ppg, length).cpu().numpy().astype(np.int64)
    return ops.calculate_bin_sizes_from_hist(hist)


def sample_from_multivariate_normal(means, covariances, num_points_to_sample, max_num_gen_attempts=3, epsilon=1e-6,
                                    seed=None, attempt=0):
    """gauss_to_pc.py:140-155 -- [n, G, 3] samples (sample-major, like MultivariateNormal.sample((n,))).
    A non positive-definite covariance shows up as NaN; like the reference, the covariances are then
    regularised in place (+epsilon I) and the draw retried, None after max_num_gen_attempts failures."""
    seed = SAMPLER_SEED if seed is None else seed
    for _ in range(max_num_gen_attempts):
        out = ops.sample_mvn(means, covariances, num_points_to_sample, seed, 0, attempt)
        if not bool(torch.isnan(out).any()):
            return out
        covariances += (epsilon * torch.eye(3, device=covariances.device, dtype=covariances.dtype))
    return None


def _finish_dtypes(points, colours, normals):
    if REFERENCE_DTYPES:
        colours = colours.to(torch.double)
        normals = normals.to(torch.double) if normals is not None else None
    return points, colours, normals


def create_new_gaussian_points(num_points_to_sample, means, covariances, colours, mahalanobis_distance_std=2,
                               num_attempts=5, normals=None, max_num_gen_attemps=3, device="cuda:0", seed=None):
    """gauss_to_pc.py:157-275 -- `num_points_to_sample` Mahalanobis-bounded points per Gaussian (first-k
    emission quirk included), one set of launches for all Gaussians."""
    seed = SAMPLER_SEED if seed is None else seed
    n = int(num_points_to_sample)
    G = means.shape[0]
    ppg = torch.full((G,), n + 1, dtype=torch.int32, device=means.device)
    out = ops.sample_pointcloud(means, covariances, colours, normals, ppg, n + 1, exact=True,
                                std=mahalanobis_distance_std, attempts=num_attempts, seed=seed,
                                bins=[(float(n + 1), float(n + 2), n + 1)], emit_means=False)
    return _finish_dtypes(out.points, out.colours, out.normals)


def generate_pointcloud(gaussians, num_points, contributions=None, mahalanobis_distance_std=2, exact_num_points=False,
                        calculate_normals=True, num_sample_attempts=5, device="cuda:0", quiet=False, seed=None,
                        gid_base=0, shard=None):
    """gauss_to_pc.py:277-371.  `shard=(rank, world)`: the allocation (sizes, points per Gaussian, bins) is computed
    for ALL Gaussians -- identical on every rank -- and only the Gaussians of this rank's contiguous index range are
    sampled; the union over ranks is exactly the single-GPU cloud (noise is keyed by the global Gaussian index)."""
    seed = SAMPLER_SEED if seed is None else seed

    # Calculate Gaussian sizes
    gaussian_sizes = gaussians.get_gaussian_magnitudes(contributions=contributions)

    if not quiet:
        print(f"Distributed Points to Gaussians")
        print()

    # Assign points to gaussians
    _, points_per_gaussian, stats = ops.distribute_points(gaussian_sizes, num_points)
    max_ppg = None                      # read back together with the histogram (one host round trip, ops.sample_pointcloud)

    bins = None
    if shard is not None and shard[1] > 1:
        max_ppg = int(stats[3].item())
        rank, world = shard
        g_total = points_per_gaussian.shape[0]
        lo, hi = (g_total * rank) // world, (g_total * (rank + 1)) // world
        hist = ops.bincount(points_per_gaussian, max_ppg + 1).cpu().numpy().astype(np.int64)
        bins = ops.bin_table_from_hist(hist, bool(exact_num_points))          # global bin table
        local = torch.full_like(points_per_gaussian, -1)                      # -1: not a member of any bin
        local[lo:hi] = points_per_gaussian[lo:hi]
        points_per_gaussian = local

    if not quiet:
        print(f"Starting Point Cloud Generation")

    out = ops.sample_pointcloud(gaussians.xyz, gaussians.covariances, gaussians.colours,
                                gaussians.normals if calculate_normals else None, points_per_gaussian, max_ppg,
                                exact=bool(exact_num_points), std=mahalanobis_distance_std,
                                attempts=num_sample_attempts, seed=seed, gid_base=gid_base, bins=bins, stats=stats)
    return _finish_dtypes(out.points, out.colours, out.normals)


DEFER_VALIDATE_CULL = True      # validate_covariances' culled-row count is read after the sampling was queued, not before


def convert_gaussians_to_pc(gaussians, transforms, intrinsics, mask_images, pointcloud_settings, seed=None,
                            group=None, render_shs=False, keep_render_context=True, single_process=False,
                            stage_times=None):
    """The body of convert_3dgs_to_pc (gauss_to_pc.py:414-601) on already-loaded data:
    `gaussians` is a gauss_handler.Gaussians, transforms / intrinsics are name -> 4x4 c2w / [w, h, fx, fy].
    Under torch.distributed (one process per GPU) the cameras are split over the ranks, the per-Gaussian
    visibility state is all-reduced once, and every rank returns the points of its Gaussian-index shard
    (g2pc.dist.gather_pointcloud assembles them); see g2pc/dist.py.
    render_shs=True hands gaussians.shs to the native rasteriser (SH evaluated per camera, forward.cu:22-73); the
    reference's convert_3dgs_to_pc never does (gauss_to_pc.py:429-432) and renders the DC colours.
    keep_render_context=False releases the renderer's pooled device context (scene copies, workspaces, captured camera
    graphs) before sampling -- what a one-shot conversion wants; a process converting scene after scene keeps it.
    single_process=True ignores an initialised torch.distributed (no camera split, no collective): the process warm-up's
    miniature job runs on every rank by itself (g2pc/warmup.py).
    stage_times: a dict to fill with this rank's milliseconds per stage (setup, camera_loop, exchange, fixed, sample, cameras
    rendered); the device is synchronised at every stage boundary, so only diagnostics passes ask for it (bench.py's
    `per_rank` block)."""
    from g2pc.dist import rank_world
    s = pointcloud_settings
    device = gaussians.xyz.device
    rank, world = (0, 1) if single_process else rank_world(group)

    import time as _time
    _t_last = [_time.perf_counter()]

    def _stage(name):
        if stage_times is None:
            return
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        now = _time.perf_counter()
        stage_times[name] = stage_times.get(name, 0.0) + (now - _t_last[0]) * 1e3
        _t_last[0] = now

    # Calculate Gaussian Normals
    if s.calculate_normals:
        gaussians.calculate_normals()

    total_gaussian_contributions = None

    if s.render_colours:
        if not s.quiet:
            print("Rendering Gaussian Colours")

        # fewer cameras than ranks: split every camera's TILES over the ranks instead of the cameras (both semantics: the
        # python-semantics renderer merges once after the loop, the native-rasteriser one camera by camera)
        split_tiles = (world > 1 and transforms is not None and len(transforms) < world)
        extra = dict(tile_shard=(rank, world)) if split_tiles else {}
        gaussian_renderer = get_renderer(s.renderer_type, gaussians.xyz, torch.unsqueeze(torch.clone(gaussians.opacities), 1),
                                         gaussians.colours, gaussians.covariances,
                                         shs=gaussians.shs if (render_shs and s.renderer_type != "python") else None,
                                         visible_gaussian_threshold=s.visibility_threshold,
                                         surface_distance_std=s.surface_distance_std,
                                         calculate_surface_distance=True if (s.surface_distance_std is not None or s.generate_mesh) else False,
                                         **extra)

        if split_tiles and hasattr(gaussian_renderer, "tile_group"):
            gaussian_renderer.tile_group = group

        if transforms is None:
            raise Exception("Transforms are required to render colours")

        # camera set-up for the whole rig at once (one batched inverse) -- except the masked cameras, whose
        # mask pins the render size and travels with the camera
        from camera_handler import get_cameras
        unmasked = {k: v for k, v in transforms.items() if mask_images is None or k not in mask_images}
        if world > 1 and not split_tiles:
            unmasked = {k: v for i, (k, v) in enumerate(transforms.items()) if i % world == rank and k in unmasked}
        # ... in two instalments: the first cameras now, the others when the loop first asks for one -- by then the first batch
        # is on the device and the ~0.5 ms of host work (one batched inverse, one Camera object each) hides behind it
        _names = list(unmasked)
        _first = 2 * 2                                                     # (two pipeline batches)
        rig = get_cameras(s.renderer_type, {k: unmasked[k] for k in _names[:_first]}, intrinsics,
                          colour_resolution=s.colour_resolution, sh_degree=s.max_sh_degree, white_bkgd=True)
        _later = {k: unmasked[k] for k in _names[_first:]}
        epoch = CAMERA_EPOCH
        if world > 1 and hasattr(gaussian_renderer, "seq_bits"):
            # every rank's keys must share one layout whatever cameras it renders: the widest tile field from the start
            # (16384 leaf tiles; 63 cameras per epoch instead of 255)
            gaussian_renderer.seq_bits = 14
            epoch = min(CAMERA_EPOCH, gaussian_renderer.camera_epoch)
        _stage("setup_ms")
        _rendered = 0
        for cam_index, (img_name, transform) in enumerate(transforms.items()):
            epochs = getattr(gaussian_renderer, "needs_camera_epochs", False)
            if world > 1:
                if epochs and cam_index > 0 and cam_index % epoch == 0:   # camera-order field of the keys
                    gaussian_renderer.all_reduce_visibility(group)
                    gaussian_renderer.rebase_keys()
                if not split_tiles and cam_index % world != rank:
                    continue
            camera = rig.get(img_name)
            if camera is None and img_name in _later:
                rig.update(get_cameras(s.renderer_type, _later, intrinsics, colour_resolution=s.colour_resolution,
                                       sh_degree=s.max_sh_degree, white_bkgd=True))
                _later = {}
                camera = rig.get(img_name)
            if camera is None:
                mask = mask_images[img_name].to(device)
                camera = get_camera(s.renderer_type, torch.tensor(list(transform)), intrinsics[img_name],
                                    colour_resolution=s.colour_resolution, sh_degree=s.max_sh_degree, white_bkgd=True, mask=mask)
            # Render new image and Gaussian contributions (the image itself is not used by the pipeline)
            _rendered += 1
            if world > 1:
                gaussian_renderer(camera, return_image=False, slot=(cam_index % epoch if epochs else cam_index) + 1)
            else:
                gaussian_renderer(camera, return_image=False)
        if stage_times is not None:
            gaussian_renderer.get_gaussian_colours()          # drains the cameras in flight and resolves the colours
            stage_times["cameras"] = _rendered
            # how many of them left the capture-and-replay path: children of split leaves / nodes rendered (quad-tree passes),
            # cameras rendered again because they did not fit their graph's buffers
            stage_times["split_children"] = int(getattr(gaussian_renderer, "split_leaves", 0))
            stage_times["child_pass_cameras"] = int(getattr(gaussian_renderer, "child_pass_cameras", 0))
            stage_times["host_driven_cameras"] = int(getattr(gaussian_renderer, "host_driven", 0))
            stage_times["rerendered_cameras"] = int(getattr(gaussian_renderer, "rerendered", 0))
        _stage("camera_loop_ms")
        if world > 1:
            gaussian_renderer.all_reduce_visibility(group)
        _stage("exchange_ms")

        if not s.quiet:
            print()
            print(f"Number Initial Gaussians: {gaussians.xyz.shape[0]}")

        # Get new rendered Gaussian colours
        gaussians.colours = gaussian_renderer.get_gaussian_colours()

        # Remove Gaussians that are not close to the predicted surface (depending on the STD)
        if s.surface_distance_std is not None:
            gaussians.add_gaussians_to_cull(gaussian_renderer.get_gaussians_with_low_surface_distance())

        # Remove Gaussians that were not rendered at all
        if s.remove_unrendered_gaussians:
            gaussians.add_gaussians_to_cull(gaussian_renderer.get_visible_gaussians())

        gaussians.apply_min_opacity(s.min_opacity)
        gaussians.apply_bounding_box(s.bounding_box_min, s.bounding_box_max)
        gaussians.cull_large_gaussians(s.cull_large_percentage)

        culled_indices = gaussians.filter_gaussians()

        if not s.quiet:
            print()
            print(f"Number Gaussians after Culling: {gaussians.xyz.shape[0]}")

        if gaussians.xyz.shape[0] < 1:
            raise Exception("Number of Gaussians after culling is 0, meaning a point cloud cannot be generated")

        if s.generate_mesh:
            surface_gaussian_idxs = gaussian_renderer.get_predicted_surface_gaussians(predicted_surface_std=1.0)
            surface_gaussian_idxs = gaussians.select(surface_gaussian_idxs)

        if s.prioritise_visible_gaussians:
            total_gaussian_contributions = gaussians.select(gaussian_renderer.get_total_gaussian_contributions())

        del gaussian_renderer
        if not keep_render_context:
            import gauss_render
            gauss_render.clear_context_pool()

    else:
        # Convert colours from (0-1) to (0-255)
        gaussians.colours = gaussians.colours * 255

        if not s.quiet:
            print("Skipping Rendering Gaussian Colours")

    if not s.quiet:
        print()
        print("Ensuring Gaussians are Positive Semidefinite")

    # (the number of culled rows stays on the device: the sampling below is queued as if nothing was culled -- the usual case --
    # and the host asks afterwards, when the device has caught up anyway, instead of stalling here in the middle of the job)
    invalid_gaussian_indices = gaussians.validate_covariances(defer_cull=DEFER_VALIDATE_CULL)

    if total_gaussian_contributions is not None and gaussians.last_validate_culled:
        total_gaussian_contributions = gaussians.select(total_gaussian_contributions)      # [invalid_gaussian_indices]

    num_sample_attempts = 5 if not s.exact_num_points else 100
    _stage("fixed_ms")

    if not s.quiet:
        print()
        print("Starting Point Cloud Generation for All Gaussians")
        print()

    def _generate():
        return generate_pointcloud(gaussians, s.num_points, exact_num_points=s.exact_num_points,
                                   mahalanobis_distance_std=s.mahalanobis_distance_std,
                                   calculate_normals=s.calculate_normals,
                                   num_sample_attempts=num_sample_attempts,
                                   contributions=total_gaussian_contributions,
                                   device=s.device, quiet=s.quiet, seed=seed,
                                   shard=(rank, world))

    early_error = None
    try:
        points, colours, normals = _generate()
    except Exception as e:          # noqa: BLE001  (judged below: it may stem from rows the deferred cull removes)
        early_error = e
    if gaussians.resolve_deferred_cull():
        # rows WERE culled (ill-conditioned covariances): whatever came out above came from the unfiltered set -- sample again
        if total_gaussian_contributions is not None:
            total_gaussian_contributions = gaussians.select(total_gaussian_contributions)  # [invalid_gaussian_indices]
        points, colours, normals = _generate()
    elif early_error is not None:
        raise early_error

    total_point_cloud = PointCloudData(points=points, colours=colours, normals=normals)
    _stage("sample_ms")

    surface_point_cloud = None

    # Generate surface point cloud if meshing the scene (gauss_to_pc.py:563-592)
    if s.generate_mesh and s.render_colours:
        if not s.quiet:
            print("Starting Point Cloud Generation for Surface Gaussians")
            print()

        if gaussians.last_validate_culled:
            surface_gaussian_idxs = gaussians.select(surface_gaussian_idxs)                  # [invalid_gaussian_indices]
        gaussians.add_gaussians_to_cull(surface_gaussian_idxs)
        gaussians.filter_gaussians()

        avg_points_per_gauss_for_mesh = 25
        total_mesh_points = min(s.num_points // 2, int(gaussians.xyz.shape[0] * avg_points_per_gauss_for_mesh))

        points, colours, normals = generate_pointcloud(gaussians, total_mesh_points, exact_num_points=s.exact_num_points,
                                                       num_sample_attempts=num_sample_attempts,
                                                       contributions=gaussians.select(total_gaussian_contributions),
                                                       device=s.device, quiet=s.quiet, seed=seed, shard=(rank, world))

        surface_point_cloud = PointCloudData(points=points, colours=colours, normals=normals)

    return total_point_cloud, surface_point_cloud


def _warmup_semantics(s):
    return () if not s.render_colours else (("python",) if s.renderer_type == "python" else ("cuda",))


def convert_3dgs_to_pc(input_path, transform_path, mask_path, pointcloud_settings):
    """gauss_to_pc.py:373-601.  File loading (gauss_dataloader.py, transform_dataloader.py, mask_dataloader.py) is the
    I/O layer around the hot path; the compute body is convert_gaussians_to_pc."""
    s = pointcloud_settings
    from transform_dataloader import load_transform_data
    from gauss_dataloader import load_gaussians

    transforms = intrinsics = mask_images = None
    if transform_path is not None:
        if not s.quiet:
            print("Loading Camera Poses")
            print()
        transforms, intrinsics = load_transform_data(transform_path, skip_rate=s.camera_skip_rate)

    if mask_path is not None:
        from mask_dataloader import load_image_masks
        if not s.quiet:
            print("Loading Masks")
            print()
        mask_images = load_image_masks(mask_path)
        for mask_name in mask_images.keys():
            if mask_name not in transforms.keys():
                print(f"WARNING: Mask with name {mask_name} not found in provided transforms")

    if not s.quiet:
        print("Loading Gaussians from File")
        print()

    xyz, scales, rots, colours, opacities, shs = load_gaussians(input_path, max_sh_degree=s.max_sh_degree)
    gaussians = Gaussians(xyz, scales, rots, colours, opacities, shs=shs)

    if torch.cuda.is_available() and str(s.device).startswith("cuda"):
        from g2pc.warmup import warmup
        warmup(s.device, _warmup_semantics(s))     # returns at once when main()'s background warm-up has finished, else waits for it

    out = convert_gaussians_to_pc(gaussians, transforms, intrinsics, mask_images, s, keep_render_context=False)

    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    gc.collect()
    return out


def config_parser(argv=None):
    """gauss_to_pc.py:603-710 -- identical flags, defaults and validation (argparse instead of configargparse)."""
    parser = argparse.ArgumentParser()

    parser.add_argument("--input_path", type=str, required=True, help="Path to ply or splat file to convert to a point cloud")
    parser.add_argument("--output_path", type=str, default="3dgs_pc.ply", help="Path to output file (must be ply file)")
    parser.add_argument("--transform_path", default=None, type=str, help="Path to COLMAP or Transform file used for loading in camera positions for rendering")
    parser.add_argument("--mask_path", default=None, type=str, help="Path to directory containing associated masks for image transforms")
    parser.add_argument("--renderer_type", type=str, default="cuda", help="'cuda' (the native tile rasteriser -- here the HIP one, alias 'hip') or 'python'")
    parser.add_argument("--num_points", type=int, default=10000000, help="Total number of points to generate for the pointcloud")
    parser.add_argument("--exact_num_points", action="store_true", help="Set if the number of generated points should more closely match the num_points argument (slower)")
    parser.add_argument("--no_prioritise_visible_gaussians", action="store_true", help="Gaussians that contribute most to the scene are given more points- set to turn this off")
    parser.add_argument("--visibility_threshold", type=float, default=0.05, help="Minimum contribution each Gaussian must have to be included in the final point cloud generation")
    parser.add_argument("--surface_distance_std", type=float, default=None, help="Cull Gaussians that are a minimum of X standard deviations away from the scene surfaces")
    parser.add_argument("--clean_pointcloud", action="store_true", help="Set to remove outliers on the point cloud after generation (requires Open3D)")
    parser.add_argument("--generate_mesh", action="store_true", help="Set to also generate a mesh based on the created point cloud (requires Open3D)")
    parser.add_argument("--poisson_depth", default=10, type=int, help="The depth used in the poisson surface reconstruction algorithm")
    parser.add_argument("--laplacian_iterations", default=10, type=int, help="The number of iterations to perform laplacian mesh smoothing")
    parser.add_argument("--mesh_output_path", type=str, default="3dgs_mesh.ply", help="Path to mesh output file (must be ply file)")
    parser.add_argument("--camera_skip_rate", type=int, default=0, help="Number of cameras to skip for each rendered camera")
    parser.add_argument("--no_render_colours", action="store_true", help="Skip rendering colours- faster but colours will be strange")
    parser.add_argument("--colour_quality", type=str, default="high", help="tiny, low, medium, high, ultra or original")
    parser.add_argument("--bounding_box_min", nargs=3, help="Values for minimum position of gaussians to include")
    parser.add_argument("--bounding_box_max", nargs=3, help="Values for maximum position of gaussians to include")
    parser.add_argument("--mahalanobis_distance_std", type=float, default=2.0, help="Maximum distance each point can be from the centre of their gaussian")
    parser.add_argument("--no_calculate_normals", action="store_true", help="Set to not calculate normals for the points")
    parser.add_argument("--min_opacity", type=float, default=0.0, help="Minimum opacity for gaussians to be included (must be between 0-1)")
    parser.add_argument("--cull_gaussian_sizes", type=float, default=0.0, help="The percentage of gaussians to remove from largest to smallest")
    parser.add_argument("--max_sh_degree", type=int, default=3, help="The number spherical harmonics of the loaded point cloud")
    parser.add_argument("--quiet", action="store_true", help="Set to surpress any output print statements")

    args = parser.parse_args(argv)

    if args.min_opacity < 0 or args.min_opacity > 1:
        raise AttributeError("Minumum opacity must be between 0 and 1")

    if args.mahalanobis_distance_std <= 0:
        raise AttributeError("Std distance must be greater than 0")

    if args.num_points <= 0:
        raise AttributeError("Number of points must be greater than 0")

    if args.bounding_box_min is not None:
        try:
            args.bounding_box_min = [float(x) for x in args.bounding_box_min]
        except ValueError:
            raise AttributeError("Bounding Box Min must contain float values")
        if len(args.bounding_box_min) != 3:
            raise AttributeError("Bounding Box Min must have exactly 3 values")

    if args.bounding_box_max is not None:
        try:
            args.bounding_box_max = [float(x) for x in args.bounding_box_max]
        except ValueError:
            raise AttributeError("Bounding Box Max must contain float values")
        if len(args.bounding_box_max) != 3:
            raise AttributeError("Bounding Box Max must have exactly 3 values")

    if args.colour_quality.lower() not in COLOR_QUALITY_OPTIONS.keys():
        raise AttributeError(f"Colour quality must be in the following options {COLOR_QUALITY_OPTIONS.keys()}")

    if args.max_sh_degree < 0:
        raise AttributeError(f"The number of spherical harmonics must be larger than 0")

    if args.camera_skip_rate < 0:
        raise AttributeError(f"The camera skip rate must be larger than 0")

    if args.generate_mesh:
        from mesh_handler import open3d_available
        if not open3d_available():
            raise AttributeError("--generate_mesh needs Open3D (Poisson surface reconstruction), which is not installed: "
                                 "refusing before the render / sampling work is spent")
    if args.generate_mesh and args.no_calculate_normals:
        raise AttributeError(f"Normals are required for accurate meshing")

    if args.generate_mesh and args.no_render_colours:
        raise AttributeError(f"Colours are required for meshing")

    if args.generate_mesh and args.transform_path is None:
        raise AttributeError(f"Transforms are required for meshing")

    if not args.no_render_colours and args.transform_path is None:
        raise AttributeError(f"Transforms are required for rendering accurate point colours, set --no_render_colours to True to render with no colour")

    if args.visibility_threshold < 0.0 or args.visibility_threshold > 1.0:
        raise AttributeError(f"Visible Gaussian Threshold must be between 0 and 1")

    if args.surface_distance_std is not None and args.surface_distance_std <= 0.0:
        raise AttributeError("Surface std must be large than 0")

    if args.mask_path is not None and args.transform_path is None:
        raise AttributeError("Cannot use masks when no transforms have been provided")

    if args.renderer_type not in ("cuda", "hip") and args.surface_distance_std is not None:
        raise AttributeError("Surface distance calculations only supported in CUDA renderer")

    return args


def settings_from_args(args):
    """gauss_to_pc.py:716-737."""
    return GaussPointCloudSettings(
        renderer_type=args.renderer_type,
        num_points=args.num_points,
        prioritise_visible_gaussians=not args.no_prioritise_visible_gaussians,
        mahalanobis_distance_std=args.mahalanobis_distance_std,
        camera_skip_rate=args.camera_skip_rate,
        render_colours=not args.no_render_colours,
        min_opacity=args.min_opacity,
        bounding_box_min=args.bounding_box_min,
        bounding_box_max=args.bounding_box_max,
        calculate_normals=not args.no_calculate_normals,
        cull_large_percentage=args.cull_gaussian_sizes,
        colour_resolution=COLOR_QUALITY_OPTIONS[args.colour_quality.lower()],
        max_sh_degree=args.max_sh_degree,
        exact_num_points=args.exact_num_points,
        generate_mesh=args.generate_mesh,
        visibility_threshold=args.visibility_threshold,
        surface_distance_std=args.surface_distance_std,
        quiet=args.quiet,
        remove_unrendered_gaussians=True if args.visibility_threshold > 0 else False,
        device=("cuda:%d" % torch.cuda.current_device()) if torch.cuda.is_available() else "cpu",
    )


def main(argv=None):
    """gauss_to_pc.py:712-786."""
    args = config_parser(argv)

    # one process per GPU (python -m torch.distributed.run --nproc-per-node N gauss_to_pc.py ...): cameras and sampling
    # are sharded over the ranks (g2pc/dist.py), rank 0 assembles and writes the cloud
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
        rank = dist.get_rank()
        args.quiet = args.quiet or rank != 0

    pointcloud_settings = settings_from_args(args)

    # load every kernel / torch operator of the hot path on a thread while the host parses the input files (g2pc/warmup.py)
    if torch.cuda.is_available() and str(pointcloud_settings.device).startswith("cuda"):
        from g2pc.warmup import warmup_in_background
        warmup_in_background(pointcloud_settings.device, _warmup_semantics(pointcloud_settings))

    total_point_cloud, surface_point_cloud = convert_3dgs_to_pc(args.input_path, args.transform_path, args.mask_path,
                                                                pointcloud_settings)

    if world > 1:
        from g2pc.dist import gather_pointcloud
        total_point_cloud = gather_pointcloud(total_point_cloud, dst=0)
        if surface_point_cloud is not None:
            surface_point_cloud = gather_pointcloud(surface_point_cloud, dst=0)
        if rank != 0:
            dist.barrier()
            dist.destroy_process_group()
            return

    if args.clean_pointcloud:
        if not args.quiet:
            print("Cleaning Point Cloud")
            print()
        from mesh_handler import clean_point_cloud      # Open3D post-process: the reference's module, out of scope
        cleaned = clean_point_cloud(total_point_cloud.points, total_point_cloud.colours, total_point_cloud.normals,
                                    device=pointcloud_settings.device)
        total_point_cloud = PointCloudData(*cleaned)

    if not args.quiet:
        print("Saving Final Point Cloud")

    from gauss_dataloader import save_xyz_to_ply
    save_xyz_to_ply(total_point_cloud.points, args.output_path, rgb_colors=total_point_cloud.colours,
                    normals_points=total_point_cloud.normals, chunk_size=10 ** 6, quiet=args.quiet)

    if pointcloud_settings.generate_mesh:
        if not args.quiet:
            print("Generating Mesh")
        from mesh_handler import generate_mesh
        generate_mesh(surface_point_cloud.points, surface_point_cloud.colours, surface_point_cloud.normals,
                      args.mesh_output_path, depth=args.poisson_depth, laplacian_iters=args.laplacian_iterations)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
