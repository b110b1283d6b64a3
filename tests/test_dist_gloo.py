"""world_size-2 run of the distributed pipeline on CPU (gloo + the fiber-emulated kernels): camera sharding,
all-reduce of the visibility state, Gaussian-index sharded sampling and the point gather must reproduce the
single-process result exactly (same point multiset, same per-Gaussian state)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _settings(num_points, renderer="python"):
    from gauss_to_pc import GaussPointCloudSettings
    return GaussPointCloudSettings(
        renderer_type=renderer, num_points=num_points, prioritise_visible_gaussians=True, mahalanobis_distance_std=2.0,
        camera_skip_rate=0, render_colours=True, min_opacity=0.0, bounding_box_min=None, bounding_box_max=None,
        calculate_normals=True, cull_large_percentage=0.0, remove_unrendered_gaussians=True, colour_resolution=None,
        max_sh_degree=3, exact_num_points=False, visibility_threshold=0.05,
        surface_distance_std=2.0 if renderer == "cuda" else None, generate_mesh=False, quiet=True, device="cpu")


def _run(rank, world, port, out_dir, renderer="python", epoch=255, ncam=3, warm_rank=None, pipelined=False, tile_limit=None):
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from g2pc import _native as nv
    nv._inject_for_tests(os.path.join(HERE, "hipemu", "libg2pc_emu.so"))
    from g2pc.synth import make_scene, make_cameras
    from g2pc.dist import gather_pointcloud
    from gauss_handler import Gaussians
    import gauss_to_pc
    from gauss_to_pc import convert_gaussians_to_pc
    gauss_to_pc.CAMERA_EPOCH = epoch
    import gauss_render
    saved = (gauss_render.PIPELINE_IN_EMULATOR, gauss_render.PIPELINE_STREAMS, gauss_render.CAMERA_BATCH,
             gauss_render.GaussHipRenderer.MAX_GAUSSIANS_PER_TILE, gauss_render.GaussHipRenderer.MAX_TILE_SIZE)
    if pipelined or tile_limit:
        gauss_render.PIPELINE_IN_EMULATOR = bool(pipelined)          # the capture-and-replay camera pipeline, through the emulator
        gauss_render.PIPELINE_STREAMS, gauss_render.CAMERA_BATCH = 2, 2
        if tile_limit:
            gauss_render.GaussHipRenderer.MAX_GAUSSIANS_PER_TILE = tile_limit
            gauss_render.GaussHipRenderer.MAX_TILE_SIZE = 16
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if warm_rank is not None and rank == warm_rank:
        # the process warm-up on ONE rank only, on a side thread as the CLI starts it: were its miniature job to split cameras
        # or all-reduce (it did until round 4: group=None picked up the initialised world), this rank would wait for a
        # collective the other never enters
        from g2pc import warmup as wu
        os.environ["G2PC_WARMUP_GAUSSIANS"] = "96"
        t = wu.warmup_in_background("cpu", (renderer,))
        t.join(timeout=240)
        assert not t.is_alive() and wu._DONE.get("cpu")
    sc = make_scene(1200, 91, scale_lo=0.01, scale_hi=0.06)
    transforms, intr = make_cameras(ncam, width=180, height=101, focal=155.0)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    cloud, _ = convert_gaussians_to_pc(G, transforms, intr, None, _settings(12000, renderer), seed=5)
    full = gather_pointcloud(cloud, dst=0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "%s_w%d_e%d%s.npz" % (renderer, world, epoch, "_p" if pipelined else "")), points=full.points.numpy(),
                 colours=full.colours.numpy(), normals=full.normals.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # (world 1 runs inside the pytest process: leave the module as it was found)
    (gauss_render.PIPELINE_IN_EMULATOR, gauss_render.PIPELINE_STREAMS, gauss_render.CAMERA_BATCH,
     gauss_render.GaussHipRenderer.MAX_GAUSSIANS_PER_TILE, gauss_render.GaussHipRenderer.MAX_TILE_SIZE) = saved
    gauss_render.clear_context_pool()


def test_two_rank_pipeline_equals_single_process(tmp_path):
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path))
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "python_w1_e255.npz"), np.load(tmp_path / "python_w2_e255.npz")
    assert a["points"].shape == b["points"].shape and a["points"].shape[0] > 10000

    def canon(d):
        rows = np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
        return rows[np.lexsort(rows.T[::-1])]
    assert np.array_equal(canon(a), canon(b))          # identical multiset of (xyz, rgb, normal) rows


def test_two_rank_graph_pipeline_with_overloaded_leaves(tmp_path):
    """The capture-and-replay camera pipeline under torch.distributed (what a multi-GPU job runs; the other tests of this file
    take the two-call path, the emulator's default): 7 cameras dealt over 2 ranks, leaves over a low max_gaussians_per_tile so
    that every rank's cameras go through the on-demand child pass (pass B at the flush before the visibility exchange) with
    caller-assigned camera slots and the 14-bit tile field of multi-rank jobs.  The gathered cloud is the cloud of ONE process
    on the two-call path with the same limits."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "python", 255, 7, None, False, 40)
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path), "python", 255, 7, None, True, 40), nprocs=2, join=True)
    a, b = np.load(tmp_path / "python_w1_e255.npz"), np.load(tmp_path / "python_w2_e255_p.npz")
    assert a["points"].shape == b["points"].shape and a["points"].shape[0] > 10000

    def canon(d):
        rows = np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
        return rows[np.lexsort(rows.T[::-1])]
    assert np.array_equal(canon(a), canon(b))


@pytest.mark.parametrize("ncam", [5, 19])
def test_eight_rank_pipeline_equals_single_process(tmp_path, ncam):
    """BASELINE configs[3]'s shape on the node it is meant for: 8 ranks.  ncam = 5 < 8 ranks: every rank renders every camera
    and blends its share of the tiles; ncam = 19 > 8: cameras dealt over the ranks (some ranks get 3, some 2), visibility
    all-reduced once, sampling sharded by Gaussian index into 8 ranges.  The gathered cloud is the single-process cloud."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "python", 255, ncam)
    port = 37500 + (os.getpid() % 2000) + ncam
    mp.spawn(_run, args=(8, port, str(tmp_path), "python", 255, ncam), nprocs=8, join=True)
    a, b = np.load(tmp_path / "python_w1_e255.npz"), np.load(tmp_path / "python_w8_e255.npz")
    assert a["points"].shape == b["points"].shape and a["points"].shape[0] > 10000

    def canon(d):
        rows = np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
        return rows[np.lexsort(rows.T[::-1])]
    assert np.array_equal(canon(a), canon(b))


def test_warm_up_under_torch_distributed_is_a_single_process_job(tmp_path):
    """g2pc.warmup runs its miniature as a single-process job whatever torch.distributed says."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path))
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path), "python", 255, 3, 1), nprocs=2, join=True)
    a, b = np.load(tmp_path / "python_w1_e255.npz"), np.load(tmp_path / "python_w2_e255.npz")
    rows = lambda d: np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
    ca, cb = rows(a), rows(b)
    assert ca.shape == cb.shape and np.array_equal(ca[np.lexsort(ca.T[::-1])], cb[np.lexsort(cb.T[::-1])])


def test_warm_up_failures_do_not_fail_the_job(monkeypatch, capsys):
    from g2pc import warmup as wu

    def boom(device, semantics):
        raise RuntimeError("no such kernel")
    monkeypatch.setattr(wu, "_run", boom)
    monkeypatch.setattr(wu, "_DONE", {})
    assert wu.warmup("cpu") >= 0.0 and wu._DONE.get("cpu")
    assert "warm-up failed" in capsys.readouterr().err


def test_two_rank_pipeline_cuda_semantics(tmp_path):
    """Native-rasteriser semantics: max / min / winner colour combine exactly (earliest global camera wins ties); the
    total contribution is a float SUM whose order differs across ranks, so the allocation may move by a point."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "cuda")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path), "cuda"), nprocs=2, join=True)
    a, b = np.load(tmp_path / "cuda_w1_e255.npz"), np.load(tmp_path / "cuda_w2_e255.npz")
    assert abs(a["points"].shape[0] - b["points"].shape[0]) <= 12
    if a["points"].shape == b["points"].shape:
        rows = lambda d: np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
        ca, cb = rows(a), rows(b)
        ca, cb = ca[np.lexsort(ca.T[::-1])], cb[np.lexsort(cb.T[::-1])]
        assert float((np.abs(ca - cb) > 1e-5).any(axis=1).mean()) < 0.01


@pytest.mark.parametrize("ncam", [5, 11])
def test_eight_rank_pipeline_cuda_semantics(tmp_path, ncam):
    """The native-rasteriser semantics on the 8-rank node shape.  ncam = 5 < 8 ranks: every rank renders every camera, blends
    the 16x16 tiles `rank, rank + 8, ...` and the ranks merge EACH camera before its running-state update (MAX of the packed
    keys, MIN of the surface distances, SUM of the images) -- the running SUM of contributions then sees the camera's global
    maximum on every rank.  ncam = 11 > 8: cameras dealt over the ranks, one exchange at the end (total contribution = a float
    SUM whose order differs from the single process: the allocation may move by a point or two)."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "cuda", 255, ncam)
    port = 39500 + (os.getpid() % 2000) + ncam
    mp.spawn(_run, args=(8, port, str(tmp_path), "cuda", 255, ncam), nprocs=8, join=True)
    a, b = np.load(tmp_path / "cuda_w1_e255.npz"), np.load(tmp_path / "cuda_w8_e255.npz")
    assert abs(a["points"].shape[0] - b["points"].shape[0]) <= 16 and a["points"].shape[0] > 10000
    if a["points"].shape == b["points"].shape:
        rows = lambda d: np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
        ca, cb = rows(a), rows(b)
        ca, cb = ca[np.lexsort(ca.T[::-1])], cb[np.lexsort(cb.T[::-1])]
        assert float((np.abs(ca - cb) > 1e-5).any(axis=1).mean()) < 0.01
        if ncam == 5:            # tile split: per-camera merges are exact (MAX / MIN / one writer per pixel): the same cloud
            assert np.array_equal(ca, cb)


def test_two_rank_pipeline_across_camera_epochs(tmp_path):
    """More cameras than the 8-bit camera-order field holds (epoch shrunk to 2 for the test): the ranks all-reduce and
    rebase the keys at every epoch boundary; the result must still equal the single-process run."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "python", 255, 5)
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path), "python", 2, 5), nprocs=2, join=True)
    a, b = np.load(tmp_path / "python_w1_e255.npz"), np.load(tmp_path / "python_w2_e2.npz")
    rows = lambda d: np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
    ca, cb = rows(a), rows(b)
    assert ca.shape == cb.shape
    assert np.array_equal(ca[np.lexsort(ca.T[::-1])], cb[np.lexsort(cb.T[::-1])])


def _cli_rank(rank, world, port, work_dir):
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2pc import _native as nv
    nv._inject_for_tests(os.path.join(HERE, "hipemu", "libg2pc_emu.so"))
    import gauss_to_pc
    out = "pc_w%d.ply" % world
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world))
    else:
        os.environ.pop("WORLD_SIZE", None)
    gauss_to_pc.SAMPLER_SEED = 11
    gauss_to_pc.main(["--input_path", os.path.join(work_dir, "scene.ply"), "--transform_path",
                      os.path.join(work_dir, "transforms.json"), "--renderer_type", "python", "--num_points", "15000",
                      "--colour_quality", "original", "--output_path", os.path.join(work_dir, out), "--quiet"])


def test_two_rank_cli_writes_the_single_process_cloud(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 gauss_to_pc.py ...`: rank 0 gathers and writes the cloud."""
    import json
    from emu_util import build_emu
    build_emu()
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2pc.synth import make_scene, make_cameras
    from test_emu_io_cli import _write_3dgs_ply
    import gauss_dataloader as gd
    sc = make_scene(900, 8, scale_lo=0.01, scale_hi=0.06)
    _write_3dgs_ply(tmp_path / "scene.ply", sc)
    tr, intr = make_cameras(3, width=180, height=101, focal=155.0)
    frames = [{"file_path": "%s.png" % k, "transform_matrix": tr[k]} for k in tr]
    (tmp_path / "transforms.json").write_text(json.dumps({"w": 180, "h": 101, "fl_x": 155.0, "frames": frames}))
    mp.spawn(_cli_rank, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_cli_rank, args=(2, 35500 + (os.getpid() % 2000), str(tmp_path)), nprocs=2, join=True)
    a, b = gd.read_ply_vertices(str(tmp_path / "pc_w1.ply")), gd.read_ply_vertices(str(tmp_path / "pc_w2.ply"))
    assert len(a) == len(b) and len(a) > 10000
    assert np.array_equal(np.sort(a, order=list(a.dtype.names)), np.sort(b, order=list(b.dtype.names)))


def test_two_ranks_one_camera_split_the_tiles(tmp_path):
    """Fewer cameras than ranks: every rank renders the camera but blends only its share of the tiles; the
    usual visibility exchange must reproduce the single-process result exactly."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "python", 255, 1)
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path), "python", 255, 1), nprocs=2, join=True)
    a, b = np.load(tmp_path / "python_w1_e255.npz"), np.load(tmp_path / "python_w2_e255.npz")
    rows = lambda d: np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
    ca, cb = rows(a), rows(b)
    assert ca.shape == cb.shape and ca.shape[0] > 5000
    assert np.array_equal(ca[np.lexsort(ca.T[::-1])], cb[np.lexsort(cb.T[::-1])])


def test_two_ranks_one_camera_split_the_tiles_cuda_semantics(tmp_path):
    """The same for the native-rasteriser semantics: the ranks blend alternate 16x16 tiles of the single camera and merge
    the camera's keys / surface distances / image before the running-state update, so every rank ends with the
    single-process state exactly (MAX / MIN / one-writer SUM are order-free)."""
    from emu_util import build_emu
    build_emu()
    _run(0, 1, 0, str(tmp_path), "cuda", 255, 1)
    port = 39500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, str(tmp_path), "cuda", 255, 1), nprocs=2, join=True)
    a, b = np.load(tmp_path / "cuda_w1_e255.npz"), np.load(tmp_path / "cuda_w2_e255.npz")
    rows = lambda d: np.concatenate([d["points"], d["colours"], d["normals"]], axis=1)
    ca, cb = rows(a), rows(b)
    assert ca.shape == cb.shape and ca.shape[0] > 5000
    assert np.array_equal(ca[np.lexsort(ca.T[::-1])], cb[np.lexsort(cb.T[::-1])])


def test_tile_shards_partition_the_chunk_list():
    sys.path.insert(0, os.path.join(ROOT, "3dgs-to-pc_amd"))
    from g2pc import tiles
    full = tiles.python_quadtree_layout(1280, 720, 60, 2)
    whole = sorted(zip(full["chunk_tile"].tolist(), full["chunk_pix0"].tolist()))
    for world in (2, 3, 8):
        parts = [tiles.python_quadtree_layout(1280, 720, 60, 2, (r, world)) for r in range(world)]
        got = sorted(c for p in parts for c in zip(p["chunk_tile"].tolist(), p["chunk_pix0"].tolist()))
        assert got == whole                                                   # disjoint and complete
        sizes = [len(p["chunk_tile"]) for p in parts]
        assert max(sizes) - min(sizes) <= 8                                   # one tile's worth of chunks at most
        assert all(np.array_equal(p["tile_seq"], full["tile_seq"]) for p in parts)   # keys mean the same on every rank


def _run_exchange_twice(rank, world, port, out_dir, renderer):
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from g2pc import _native as nv
    nv._inject_for_tests(os.path.join(HERE, "hipemu", "libg2pc_emu.so"))
    from g2pc.synth import make_scene, make_cameras
    from gauss_handler import Gaussians
    import camera_handler
    import gauss_render
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = make_scene(800, 93, scale_lo=0.01, scale_hi=0.06)
    transforms, intr = make_cameras(4, width=120, height=68, focal=100.0)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    R = gauss_render.get_renderer(renderer, G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                  visible_gaussian_threshold=0.05, surface_distance_std=2.0 if renderer == "cuda" else None,
                                  calculate_surface_distance=renderer == "cuda")
    states = []
    for half in (0, 1):                     # two cameras, exchange, exchange AGAIN, two more cameras, exchange
        for ci, name in list(enumerate(sorted(transforms)))[2 * half:2 * half + 2]:
            if ci % world == rank:
                cam = camera_handler.get_camera(renderer, torch.tensor(transforms[name]), intr[name])
                R(cam, return_image=False, slot=ci + 1)
        R.all_reduce_visibility()
        first = (R.get_gaussian_colours().clone(), R.get_total_gaussian_contributions().clone())
        R.all_reduce_visibility()           # nothing rendered in between: must not change anything
        second = (R.get_gaussian_colours().clone(), R.get_total_gaussian_contributions().clone())
        assert torch.equal(first[0], second[0]) and torch.equal(first[1], second[1])
        states.append(second)
    if rank == 0:
        np.savez(os.path.join(out_dir, "twice_%s.npz" % renderer), colours=states[-1][0].numpy(), total=states[-1][1].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("renderer", ["python", "cuda"])
def test_visibility_exchange_is_idempotent(tmp_path, renderer):
    """all_reduce_visibility called twice in a row (and again after more cameras) equals the single-process state: one
    rank per Gaussian is elected for the colour SUM, and the running total only ships what was added since the last
    exchange."""
    from emu_util import build_emu
    build_emu()
    port = 33500 + (os.getpid() % 2000) + (7 if renderer == "cuda" else 0)
    mp.spawn(_run_exchange_twice, args=(2, port, str(tmp_path), renderer), nprocs=2, join=True)
    got = np.load(tmp_path / ("twice_%s.npz" % renderer))
    # single process, same four cameras
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2pc import _native as nv
    saved = (nv._LIB, nv._EMULATED)
    nv._inject_for_tests(os.path.join(HERE, "hipemu", "libg2pc_emu.so"))
    try:
        from g2pc.synth import make_scene, make_cameras
        from gauss_handler import Gaussians
        import camera_handler
        import gauss_render
        sc = make_scene(800, 93, scale_lo=0.01, scale_hi=0.06)
        transforms, intr = make_cameras(4, width=120, height=68, focal=100.0)
        G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
        R = gauss_render.get_renderer(renderer, G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05, surface_distance_std=2.0 if renderer == "cuda" else None,
                                      calculate_surface_distance=renderer == "cuda")
        for name in sorted(transforms):
            R(camera_handler.get_camera(renderer, torch.tensor(transforms[name]), intr[name]), return_image=False)
        assert np.array_equal(R.get_gaussian_colours().numpy(), got["colours"])
        np.testing.assert_allclose(R.get_total_gaussian_contributions().numpy(), got["total"], rtol=1e-5, atol=1e-6)
    finally:
        nv._LIB, nv._EMULATED = saved


def _run_split_on_one_rank(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from g2pc import _native as nv
    nv._inject_for_tests(os.path.join(HERE, "hipemu", "libg2pc_emu.so"))
    from g2pc.synth import make_scene, make_cameras
    from gauss_handler import Gaussians
    import camera_handler
    import gauss_render
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = make_scene(4000, 8, scale_lo=0.004, scale_hi=0.03)
    tr, intr = make_cameras(2, width=256, height=160, focal=220.0)
    G = Gaussians(sc.xyz * 0.6, sc.scales, sc.rots, sc.colours, sc.opacities)
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances)
    R.MAX_TILE_SIZE, R.MAX_GAUSSIANS_PER_TILE = 8, 8             # 1 024 leaves; > 3 072 split children push past the 12-bit tile field
    name = sorted(tr)[rank]
    if rank == 1:
        R.MAX_GAUSSIANS_PER_TILE = 1 << 30                       # rank 1's camera splits nothing: its keys stay 12 bits wide
    R(camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name]), return_image=False, slot=rank + 1)
    R.flush()
    bits_before = R.seq_bits
    R.all_reduce_visibility()
    np.savez(os.path.join(out_dir, "split_rank%d.npz" % rank), keys=R.best_key.numpy(), colours=R.get_gaussian_colours().numpy(),
             bits_before=bits_before, bits_after=R.seq_bits)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_key_width_before_the_exchange(tmp_path):
    """One rank's camera splits leaves and widens its keys' tile field (12 -> 13 bits), the other's does not: the exchange
    brings every rank to the widest field first (g2pc_raster_repack_keys) and ends in the single-process state."""
    from emu_util import build_emu
    build_emu()
    port = 41500 + (os.getpid() % 2000)
    mp.spawn(_run_split_on_one_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "split_rank0.npz"), np.load(tmp_path / "split_rank1.npz")
    assert int(a["bits_before"]) > 12 and int(b["bits_before"]) == 12 and int(a["bits_after"]) == int(b["bits_after"]) == int(a["bits_before"])
    assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["colours"], b["colours"])
    # single process, both cameras, the wide field from the start
    for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2pc import _native as nv
    saved = (nv._LIB, nv._EMULATED)
    nv._inject_for_tests(os.path.join(HERE, "hipemu", "libg2pc_emu.so"))
    try:
        from g2pc.synth import make_scene, make_cameras
        from gauss_handler import Gaussians
        import camera_handler
        import gauss_render
        sc = make_scene(4000, 8, scale_lo=0.004, scale_hi=0.03)
        tr, intr = make_cameras(2, width=256, height=160, focal=220.0)
        G = Gaussians(sc.xyz * 0.6, sc.scales, sc.rots, sc.colours, sc.opacities)
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances)
        R.MAX_TILE_SIZE, R.MAX_GAUSSIANS_PER_TILE = 8, 8
        R.seq_bits = int(a["bits_after"])
        for rank, name in enumerate(sorted(tr)):
            if rank == 1:
                R.MAX_GAUSSIANS_PER_TILE = 1 << 30
            R(camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name]), return_image=False, slot=rank + 1)
        R.flush()
        assert np.array_equal(R.best_key.numpy(), a["keys"]) and np.array_equal(R.get_gaussian_colours().numpy(), a["colours"])
    finally:
        nv._LIB, nv._EMULATED = saved
