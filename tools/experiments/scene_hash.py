"""Are g2pc.synth scenes bit-identical across hosts?  sha256 of every array of the 1 M bench scene + CPU model."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd")]
import numpy as np, torch
from g2pc.synth import make_scene
sc = make_scene(1_000_000, 1237, with_sh=True)
for k in sc._fields:
    print(k, hashlib.sha256(getattr(sc, k).numpy().tobytes()).hexdigest()[:16])
s = torch.rand((1000000, 3), generator=torch.Generator().manual_seed(5), dtype=torch.float32) * 0.018 + 0.002
lg = torch.log(s).numpy()
cr = np.log(s.numpy().astype(np.float64)).astype(np.float32)
print("torch.log vs f64-rounded log: differing", int((lg != cr).sum()), "sha", hashlib.sha256(lg.tobytes()).hexdigest()[:16], hashlib.sha256(cr.tobytes()).hexdigest()[:16])
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0], torch.__config__.parallel_info().split("\n")[0], torch.backends.cpu.get_cpu_capability())
