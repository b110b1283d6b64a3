"""GPU box: depth order of 1 M keys -- the 4-pass radix sort against the bucket pass + in-LDS sort, alone on the device
(HIP events around 20 back-to-back calls)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd")]
import numpy as np, torch
from g2pc import _native as nv
DEV = "cuda:0"
L = nv.lib()
for n in (1_000_000, 5_000_000):
    rng = np.random.default_rng(n)
    depth = rng.uniform(2.5, 4.5, size=n).astype(np.float32)
    keys = depth.view(np.uint32).copy()
    keys[rng.random(n) < 0.1] = 0xFFFFFFFF
    k = torch.from_numpy(keys.view(np.int32)).to(DEV); v = torch.arange(n, dtype=torch.int32, device=DEV)
    ko, vo, kt, vt = (torch.empty_like(k) for _ in range(4))
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    wb = L.g2pc_bucket_sort_workspace(n); wsb = torch.empty(wb, dtype=torch.uint8, device=DEV)
    wr = L.g2pc_sort_workspace(n); wsr = torch.empty(wr, dtype=torch.uint8, device=DEV)
    st = nv.stream_handle(DEV)
    def bucket():
        nv.check(L.g2pc_bucket_sort_u32(nv.ptr(k), nv.ptr(v), nv.ptr(ko), nv.ptr(vo), n, nv.ptr(flag), nv.ptr(wsb), wb, st), "bucket")
    def radix():
        nv.check(L.g2pc_sort_pairs_u32(nv.ptr(k), nv.ptr(v), nv.ptr(ko), nv.ptr(vo), nv.ptr(kt), nv.ptr(vt), n, 0, 32, nv.ptr(wsr), wr, st), "radix")
    for name, fn in (("bucket", bucket), ("radix", radix)):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        print("n %d %s: %.1f us per sort (overflow flag %d)" % (n, name, a.elapsed_time(b) * 1e3 / 20, int(flag.item())))
