"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Harness that imports the *untouched* reference (/root/reference) on a CPU-only box.
Recipe follows SURVEY.md Appendix A:
  (1) a TorchFunctionMode that maps every hard-coded "cuda" device to cpu
      (reference hard-codes devices at gauss_handler.py:13,30,50,87,98,
       gauss_render.py:196,218-225, gauss_to_pc.py:151),
  (2) stub modules for configargparse / imageio / plyfile / cv2
      (imports at gauss_to_pc.py:4,7; gauss_dataloader.py:6; transform_dataloader.py:4),
  (3) torch.cuda.mem_get_info / memory_allocated patched so that
      gauss_render.py:440-444 picks max_gaussians_per_tile=60000, max_tile_size=60.

/root/reference exists only in the authoring container: this module is used by
oracle/make_golden.py (committed generator of tests/golden/*.npz) and by the
container-only parity tests (skipped automatically when the reference is absent).
"""
import contextlib
import os
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

REFERENCE_ROOT = os.environ.get("G2PC_REFERENCE_ROOT", "/root/reference")
TILE_PIN = 60000  # -> max_gaussians_per_tile = 60000, max_tile_size = 60


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "gauss_to_pc.py"))


def _cpu(d):
    if isinstance(d, bool):
        return d
    if isinstance(d, int):
        return "cpu"
    if isinstance(d, str) and d.startswith("cuda"):
        return "cpu"
    if isinstance(d, torch.device) and d.type == "cuda":
        return torch.device("cpu")
    return d


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device") is not None:
            kwargs["device"] = _cpu(kwargs["device"])
        if func is torch.device and args and isinstance(args[0], str) and args[0].startswith("cuda"):
            return torch.device("cpu")
        if func is torch.Tensor.to and len(args) >= 2:
            args = (args[0], _cpu(args[1]), *args[2:])
        return func(*args, **kwargs)


_loaded = {}


def load_reference():
    """Import the reference's python modules (unmodified) and return them in a dict."""
    if _loaded:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference sources not present at %s" % REFERENCE_ROOT)
    torch.cuda.mem_get_info = lambda *a, **k: (TILE_PIN * 175000, TILE_PIN * 175000)
    torch.cuda.memory_allocated = lambda *a, **k: 0
    torch.cuda.empty_cache = lambda *a, **k: None
    for m in ("configargparse", "imageio", "plyfile", "cv2"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    # the reference's module names collide with the drop-in package's: make sure the
    # reference directory wins inside this (generator) process.
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k in ("gauss_to_pc", "gauss_handler", "gauss_render", "camera_handler",
                      "gauss_dataloader", "transform_dataloader", "mask_dataloader")}
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        with CudaToCpu():
            import gauss_handler, gauss_render, camera_handler, gauss_to_pc  # noqa
        _loaded.update(gauss_handler=gauss_handler, gauss_render=gauss_render,
                       camera_handler=camera_handler, gauss_to_pc=gauss_to_pc)
        # keep the reference modules reachable under private names, restore whatever was there
        for k in ("gauss_to_pc", "gauss_handler", "gauss_render", "camera_handler",
                  "gauss_dataloader", "transform_dataloader", "mask_dataloader"):
            mod = sys.modules.pop(k, None)
            if mod is not None:
                sys.modules["_ref_" + k] = mod
    finally:
        sys.path.remove(REFERENCE_ROOT)
        sys.modules.update(saved)
    return _loaded


@contextlib.contextmanager
def inject_standard_normal(provider):
    """Route torch.distributions.MultivariateNormal's noise through `provider(shape)`.

    The reference draws eps of shape [n, G_active, 3] once per (bin, attempt)
    (gauss_to_pc.py:149 -> torch/distributions/multivariate_normal.py `_standard_normal`).
    """
    import torch.distributions.multivariate_normal as mvn
    orig = mvn._standard_normal

    def fake(shape, dtype, device):
        return provider(tuple(shape)).to(dtype=dtype, device=device)

    mvn._standard_normal = fake
    try:
        yield
    finally:
        mvn._standard_normal = orig


# ---- the reference's NATIVE rasteriser, compiled for the host (oracle/build_ref.py -> oracle/_ref/) ----
GPR_NAME = "gaussian_pointcloud_rasterization"


def load_reference_gpr(variant: str = "synced", fma: bool = False):
    """The reference's own gaussian_pointcloud_rasterization/__init__.py (unmodified), bound to the `_C` extension that
    oracle/build_ref.py compiled from the reference's .cu files.  Returned as a module object that is NOT left in
    sys.modules (the product ships a drop-in package of the same name); use `reference_gpr()` around reference calls
    that import it by name (gauss_render.get_renderer / camera_handler.get_camera, "cuda" branch)."""
    import importlib.util
    import build_ref
    key = ("gpr", variant, fma)
    if key in _loaded:
        return _loaded[key]
    so_dir = build_ref.build(variant, fma)
    init = os.path.join(REFERENCE_ROOT, "gaussian-pointcloud-rasterization", GPR_NAME, "__init__.py")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == GPR_NAME or k.startswith(GPR_NAME + ".")}
    try:
        spec = importlib.util.spec_from_file_location(GPR_NAME, init, submodule_search_locations=[so_dir])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[GPR_NAME] = mod
        with CudaToCpu():
            spec.loader.exec_module(mod)          # runs `from . import _C` -> oracle/_ref/<variant>/_C.so
        assert os.path.dirname(mod._C.__file__) == so_dir, mod._C.__file__
    finally:
        for k in [k for k in sys.modules if k == GPR_NAME or k.startswith(GPR_NAME + ".")]:
            del sys.modules[k]
        sys.modules.update(saved)
    _loaded[key] = mod
    return mod


@contextlib.contextmanager
def reference_gpr(variant: str = "synced", fma: bool = False):
    """Inside the block, `import gaussian_pointcloud_rasterization` resolves to the reference's package on the host build."""
    mod = load_reference_gpr(variant, fma)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == GPR_NAME or k.startswith(GPR_NAME + ".")}
    sys.modules[GPR_NAME] = mod
    try:
        with CudaToCpu():
            yield mod
    finally:
        sys.modules.pop(GPR_NAME, None)
        sys.modules.update(saved)
