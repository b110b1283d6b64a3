"""CONTAINER ONLY (needs /root/reference and oracle/_ref) -- random native-semantics ("cuda" renderer) cases through the reference's
OWN rasteriser (its .cu sources compiled for the host, driven through its python binding: oracle/make_golden_cu.py) and through
the product (emulator), compared with the fixture checker (tests/cu_golden.py, the bars for builds whose exp differs in the last
bit: alpha / transmittance cut-offs flip at isolated pixels).  Image sizes 20 .. 500 and, a quarter of the cases, wider than 4 096
pixels; SH or precomputed colours, surface distance, masks, one or two cameras.
usage: python tools/experiments/cuda_semantics_fuzz_vs_reference.py <seed> <cases>.  Round 3: 60 cases -- radii, instance counts,
tiles touched, projected means / depths / conics bit for bit in all of them; one case one arg-max tie over the bar (3 of 11 705)."""
import sys, os, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, '3dgs-to-pc_amd')]
import numpy as np
import ref_shim
if not ref_shim.reference_available():
    sys.exit('the reference sources are not here')
import make_golden_cu as MGC
import cu_golden
ref = ref_shim.load_reference()
from g2pc import _native as nv
from emu_util import build_emu
nv._inject_for_tests(build_emu())
from cuda_checks import run_golden_case
tmp = tempfile.mkdtemp(prefix="cu_fuzz_")
MGC.GOLD = tmp; cu_golden.GOLD = tmp
# (the spread over the reference's other build variants costs three subprocesses per case: not needed here)
MGC._run_case_in_subprocess = lambda name, variant, fma: MGC.run_case(ref, name, "synced", False)[:2]
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    wide = rng.random() < 0.25
    W = int(rng.integers(4100, 4700)) if wide else int(rng.integers(20, 500)); H = int(rng.integers(17, 64)) if wide else int(rng.integers(17, 300))
    n = int(rng.integers(300, 5000)) * (5 if wide else 1)
    c = dict(n=n, seed=4000 + it, width=W, height=H, focal=0.9 * W, ncam=int(rng.integers(1, 3)), scale=(0.004, float(rng.choice([0.02, 0.05]))),
             with_sh=bool(rng.integers(0, 2)), surf=bool(rng.integers(0, 2)), mask=str(rng.choice(["none", "ones", "band_disc"])) if not wide else "none",
             pixel_stride=1, store_list=False, store_geom=True)
    name = "fuzz%d" % it
    MGC.CASES[name] = c
    t = time.time()
    try:
        MGC.generate(ref, name)
        reps, st, case = run_golden_case(name)
        for rep in reps:
            cu_golden.assert_camera(rep, case, strict=False)
        cu_golden.assert_state(st, case, strict=False)
        print(it, json.dumps({k: c[k] for k in ("n", "width", "height", "ncam", "with_sh", "surf", "mask")}), "OK %.1fs" % (time.time() - t), flush=True)
    except AssertionError as e:
        bad += 1; print(it, json.dumps(c), "MISMATCH", str(e)[:1500], flush=True)
    finally:
        try: os.remove(os.path.join(tmp, "render_cu_%s.npz" % name))
        except OSError: pass
print("mismatches", bad)
