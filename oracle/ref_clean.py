"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's point-cloud cleaning (never imported by the product).

Path: gauss_to_pc.py:743-759 -> mesh_handler.py:89-94 clean_point_cloud -> mesh_handler.py:42-64 (torch <-> Open3D
conversions) -> Open3D `geometry::PointCloud::RemoveStatisticalOutliers(nb_neighbors=20, std_ratio=10)`.

PARITY UNPINNED: Open3D is a third-party dependency that is neither vendored under /root/reference nor installed in
this image, and the reference pins no version (README: "pip install open3d").  The algorithm below is the published one
(Open3D 0.13 .. 0.19, cpp/open3d/geometry/PointCloud.cpp, unchanged across those releases):

    for every point i:  dist2 = squared distances to the nb_neighbors nearest points (KDTreeFlann::SearchKNN on the
                        cloud itself, so the point is its own first neighbour at distance 0), ascending;
                        avg[i] = sum(sqrt(dist2)) / len(dist2)            (float64)
    cloud_mean = sum(avg[avg > 0]) / valid          valid = number of points whose search returned something
    std_dev    = sqrt(sum((avg - cloud_mean)^2 for avg > 0) / (valid - 1))
    keep i  <=>  avg[i] > 0 and avg[i] < cloud_mean + std_ratio * std_dev

and the conversions around it are the reference's own lines: colours are clamped to [0, 255], truncated to int32,
divided by 255 (float64), and on the way back multiplied by 255 and truncated to int again (the round trip is exact for
every integer 0..255, tests/test_emu_clean.py checks it); points and normals come back as float64.
The kNN itself uses scipy's cKDTree (exact, float64).
"""
import numpy as np
from scipy.spatial import cKDTree


def knn_mean_distance(points_f32, k=20):
    """avg[i] of the restatement above; points float32 [M,3] -> float64 [M]."""
    pts = np.asarray(points_f32, dtype=np.float32).astype(np.float64)      # Vector3dVector(float32 array): exact widening
    kk = min(int(k), pts.shape[0])
    _, nn = cKDTree(pts).query(pts, k=kk)
    nn = nn.reshape(pts.shape[0], kk)
    diff = pts[:, None, :] - pts[nn]                                        # nanoflann L2_Simple: sum of squares in
    d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]   # dimension order
    d2 = np.sort(d2, axis=1)                        # ... returned in ascending order
    out = np.zeros(pts.shape[0], dtype=np.float64)
    for j in range(kk):                             # std::accumulate from 0.0, ascending
        out += np.sqrt(d2[:, j])
    return out / kk


def statistical_outlier_mask(avg, std_ratio=10.0):
    avg = np.asarray(avg, dtype=np.float64)
    pos = avg > 0
    valid = avg.shape[0]                            # every search on a non-empty cloud returns >= 1 neighbour
    cloud_mean = avg[pos].sum() / valid
    sq = ((avg[pos] - cloud_mean) ** 2).sum()
    std_dev = np.sqrt(sq / (valid - 1))
    thr = cloud_mean + std_ratio * std_dev
    return pos & (avg < thr), thr


def colour_round_trip(colours):
    """mesh_handler.py:47,52 then :60 -- returns int32 [M,3]."""
    c = np.clip(np.asarray(colours, dtype=np.float64), 0, 255).astype(np.int32)
    return (c.astype(np.float64) / 255 * 255).astype(np.int32)


def clean_point_cloud(points, colours, normals, std_ratio=10.0, k=20):
    """-> (points f64 [M',3], colours i32 [M',3], normals f64 [M',3] or None, keep mask, avg)."""
    avg = knn_mean_distance(points, k)
    keep, _ = statistical_outlier_mask(avg, std_ratio)
    pts = np.asarray(points, dtype=np.float32).astype(np.float64)[keep]
    cols = colour_round_trip(colours)[keep]
    nrm = None if normals is None else np.asarray(normals).astype(np.float64)[keep]
    return pts, cols, nrm, keep, avg
