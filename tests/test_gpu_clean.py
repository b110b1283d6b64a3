"""--clean_pointcloud on the MI355X: grid kNN kernels against the cKDTree restatement, and a full-size (10 M points)
run checked against brute force on a sample of queries."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sampled_cloud(m, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand((m, 2), generator=g) * 2 - 1
    z = 0.2 * torch.sin(3 * u[:, 0]) * torch.cos(2 * u[:, 1]) + 0.004 * torch.randn((m,), generator=g)
    pts = torch.cat([u, z[:, None]], 1)
    far = (torch.rand((12, 3), generator=g) * 2 - 1) * torch.tensor([6.0, 5.0, 7.0]) + torch.tensor([0.0, 0.0, 9.0])
    pts = torch.cat([pts, far]).to(torch.float32)
    return pts[torch.randperm(pts.shape[0], generator=g)].to(dev)


def test_knn_mean_distance_and_mask_match_kdtree():
    import mesh_handler
    import ref_clean
    pts = _sampled_cloud(300_000, 3, "cuda:0")
    avg = mesh_handler.knn_mean_distance(pts).cpu().numpy()
    ref = ref_clean.knn_mean_distance(pts.cpu().numpy())
    np.testing.assert_allclose(avg, ref, rtol=1e-14, atol=0)
    keep = mesh_handler.statistical_outlier_mask(torch.from_numpy(avg).cuda(), 10.0).cpu().numpy()
    ref_keep, thr = ref_clean.statistical_outlier_mask(ref, 10.0)
    assert np.array_equal(keep, ref_keep) and 0 < (~keep).sum() <= 12
    cols = torch.rand((pts.shape[0], 3), device="cuda:0") * 255
    p, c, n = mesh_handler.clean_point_cloud(pts, cols, None)
    assert p.shape[0] == int(ref_keep.sum()) and n is None and c.dtype == torch.int32
    assert np.array_equal(p.cpu().numpy(), pts.cpu().numpy().astype(np.float64)[ref_keep])


def test_full_size_cloud_against_brute_force_sample():
    import mesh_handler
    m = 10_000_000
    pts = _sampled_cloud(m, 4, "cuda:0")
    mesh_handler.knn_mean_distance(pts[:100_000])                       # warm-up (allocator, module load)
    torch.cuda.synchronize()
    t = time.perf_counter()
    avg = mesh_handler.knn_mean_distance(pts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("knn_mean_distance: %d points in %.1f ms (%.3g points/s)" % (pts.shape[0], dt * 1e3, pts.shape[0] / dt), mesh_handler.LAST_STATS)
    assert dt < 5.0
    q = torch.randperm(pts.shape[0], device="cuda:0")[:256]
    q = torch.cat([q, torch.topk(avg, 8).indices])                      # and the 8 loneliest points
    p64 = pts.double()
    ref = torch.empty((q.shape[0],), dtype=torch.float64, device="cuda:0")
    for i in range(0, q.shape[0], 8):
        d = p64[q[i:i + 8]][:, None, :] - p64[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        best = torch.topk(d2, 20, dim=1, largest=False).values.sort(dim=1).values
        acc = torch.zeros((best.shape[0],), dtype=torch.float64, device="cuda:0")
        for j in range(20):
            acc += best[:, j].sqrt()
        ref[i:i + 8] = acc / 20
    torch.testing.assert_close(avg[q], ref, rtol=1e-14, atol=0)
