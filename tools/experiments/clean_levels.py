"""Experiment (GPU box): level structure and timing of the outlier-removal cascade on a 10 M point cloud."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch
import mesh_handler
from test_gpu_clean import _sampled_cloud
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = _sampled_cloud(m, 4, "cuda:0")
mesh_handler.knn_mean_distance(pts[:100_000])
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    avg = mesh_handler.knn_mean_distance(pts)
    torch.cuda.synchronize()
    print("total %.1f ms" % ((time.perf_counter() - t) * 1e3), mesh_handler.LAST_STATS)
