"""
TEST INFRASTRUCTURE ONLY (oracle) -- build recipe for oracle/_ref/: THE REFERENCE'S OWN native rasteriser
(gaussian-pointcloud-rasterization: cuda_rasterizer/forward.cu, cuda_rasterizer/rasterizer_impl.cu, rasterize_points.cu,
ext.cpp), compiled for the host with g++ from the sources where they lie under /root/reference.  Container-only, like
oracle/make_golden.py: /root/reference does not exist on the GPU box; what travels are the golden vectors
(tests/golden/render_cu_*.npz) that make_golden.py produces with the module built here.  oracle/_ref/ is git-ignored
(no reference source, original or generated, enters the history).

How a .cu file gets through g++:
  * oracle/cuda_host/include/ supplies <cuda_runtime.h>, <cooperative_groups.h>, <cub/cub.cuh> ... on top of the fibre
    engine of the CPU test-suite (one fibre per GPU thread, real barriers);  glm is the reference's vendored copy.
  * `kernel <<<grid, block>>> (args)` is not C++.  The recipe rewrites the five launch statements -- and nothing else --
    to `CUEMU_LAUNCH((kernel), grid, block)(args)` while copying the two .cu files into oracle/_ref/gen/.
  * rasterize_points.cu is compiled untouched through oracle/cuda_host/wrap_rasterize_points.cpp (kCUDA -> kCPU for its
    scratch tensors), ext.cpp untouched.  The result is a real torch extension `_C`, so the reference's own python binding
    (gaussian_pointcloud_rasterization/__init__.py) runs unmodified on top of it (oracle/ref_shim.load_reference_gpr).

The reference's races (SURVEY §2.2 defect 3) under this engine.  The engine runs the blocks of a launch in blockIdx order
and the threads of a block in thread_rank order from barrier to barrier.  Every statement of renderCUDA that is a race on
hardware therefore has ONE outcome here:
  * shared-memory atomicMax + plain store of the pixel id (forward.cu:434-438): the first (lowest thread_rank = lowest
    pixel id of the tile) thread that reaches the maximum keeps the pixel;
  * cross-block compare-then-store of gauss_contributions / gauss_pixels (forward.cu:447-452): the first tile (lowest
    tile id) that reaches the maximum keeps it;
  * BUT the flush of a batch's per-Gaussian maximum (forward.cu:445-455) and of its surface distance (forward.cu:474-476)
    is separated from the loops that produce them (forward.cu:393-442, 463-471) by NO barrier: thread t flushes slot t as
    soon as ITS OWN loop is over.  On a GPU the 8 warps of a tile run concurrently, so what a flush sees is timing-
    dependent (the reference's output is not reproducible run to run); in thread_rank order it sees the contributions of
    threads 0..t only -- an artefact of the schedule that no GPU produces either.
  Two variants are therefore built:
    "verbatim" -- the launch rewrite only.  Pins everything that is private to a pixel or a Gaussian and hence race-free:
                  radii, num_rendered, out_color, out_depth, out_invdepth, (and final_T / n_contrib inside imgBuffer).
                  Its gauss_contributions are LOWER bounds, its gauss_surface_distances UPPER bounds, of any execution.
    "synced"   -- additionally `block.sync();` inserted in front of each of the two flushes (the statements the recipe
                  anchors on are asserted to occur exactly once).  This is the execution in which every thread of the tile
                  has finished the batch before its result is published, i.e. the one the code intends and the limit a GPU
                  approaches when its warps stay in step.  It pins gauss_contributions, gauss_pixels,
                  gauss_surface_distances and, through the binding, the running max / sum / min state and the colours.
  Each variant is built twice: WITHOUT floating-point contraction (-ffp-contract=off: every operation of the reference's
  expressions rounded on its own, in source order -- the one evaluation of its text that no compiler decides) and with it
  (-mfma -ffp-contract=fast; nvcc's default is -fmad=true, but WHICH products a compiler fuses is its own business: gcc
  fuses the first product of transformPoint4x3 and the last two of transformPoint4x4, tools/cu_preprocess_exactness.py).
  The goldens come from synced + no contraction; make_golden_cu.py stores how far the other three land from it -- that
  spread is how much of the reference's output is decided by its compiler and its races rather than by its source.

Usage:  python oracle/build_ref.py            (builds all four, a few minutes the first time; incremental afterwards)
        build("synced", fma=False) -> directory holding _C.so, for ref_shim.load_reference_gpr().
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("G2PC_REFERENCE", "/root/reference")
GPR = os.path.join(REF, "gaussian-pointcloud-rasterization")
OUT = os.path.join(HERE, "_ref")
SHIM = os.path.join(HERE, "cuda_host")

_LAUNCH = re.compile(r"([A-Za-z_]\w*(?:<[^<>()]*>)?)\s*<<\s*<\s*(.+?)\s*>>\s*>\s*\(")
_FLUSH_CONTRIB = "if(largest_collected_contributions[block.thread_rank()] > gauss_contributions"
_FLUSH_SURF = "if(smallest_collected_surface_distance[block.thread_rank()] < gauss_surface_distances"
_GUARD = "if (range.x + progress < range.y)"


def available() -> bool:
    return os.path.isfile(os.path.join(GPR, "cuda_rasterizer", "forward.cu"))


def _rewrite(text: str, name: str, synced: bool) -> str:
    text, n = _LAUNCH.subn(lambda m: "CUEMU_LAUNCH((%s), %s)(" % (m.group(1), m.group(2)), text)
    want = {"forward.cu": 2, "rasterizer_impl.cu": 3}[name]
    assert n == want, "%s: expected %d kernel launches, rewrote %d -- the reference changed" % (name, want, n)
    if synced and name == "forward.cu":
        for anchor in (_FLUSH_CONTRIB, _FLUSH_SURF):
            assert text.count(anchor) == 1, anchor
            at = text.index(anchor)
            guard = text.rindex(_GUARD, 0, at)
            assert at - guard < 80, "flush guard not adjacent to the flush -- the reference changed"
            text = text[:guard] + "block.sync(); /* inserted by oracle/build_ref.py (variant 'synced') */ " + text[guard:]
    return text


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout[-6000:] + "\n")
        raise RuntimeError("oracle/_ref build failed")


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _torch_flags():
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    inc = ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    link = ["-L" + libdir, "-Wl,-rpath," + libdir, "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python"]
    abi = "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)
    return inc, link, abi


def build(variant: str = "synced", fma: bool = False) -> str:
    """Returns the directory that holds the `_C` extension of the requested variant (building what is out of date)."""
    assert variant in ("verbatim", "synced")
    if not available():
        raise FileNotFoundError("reference sources not found under %s (oracle/_ref is container-only)" % GPR)
    gen, obj = os.path.join(OUT, "gen"), os.path.join(OUT, "obj")
    tag = "%s_%s" % (variant, "fma" if fma else "nofma")
    dst = os.path.join(OUT, tag)
    for d in (gen, obj, dst):
        os.makedirs(d, exist_ok=True)
    shim_deps = [os.path.join(dp, f) for dp, _, fs in os.walk(SHIM) for f in fs] + [
        os.path.join(HERE, "..", "tests", "hipemu", "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    inc, link, abi = _torch_flags()
    common = ["g++", "-std=c++17", "-O2", "-fPIC", "-fvisibility=hidden", "-w", "-I" + os.path.join(SHIM, "include"),
              "-I" + os.path.join(GPR, "cuda_rasterizer"), "-I" + GPR, "-I" + os.path.join(GPR, "third_party", "glm")]
    fpflags = ["-mfma", "-ffp-contract=fast"] if fma else ["-ffp-contract=off"]
    objs = []
    # (1) the two kernel files: launch rewrite (+ the two barriers of "synced"), one object per variant
    for name in ("forward.cu", "rasterizer_impl.cu"):
        src = os.path.join(GPR, "cuda_rasterizer", name)
        text = _rewrite(open(src).read(), name, synced=(variant == "synced"))
        g = os.path.join(gen, "%s.%s.cpp" % (name[:-3], variant))
        if not os.path.isfile(g) or open(g).read() != text:
            open(g, "w").write(text)
        o = os.path.join(obj, "%s.%s.o" % (name[:-3], tag))
        if _stale(o, [g] + shim_deps):
            _run(common + fpflags + ["-c", g, "-o", o])
        objs.append(o)
    # (2) the torch binding, untouched (shared by the variants)
    rp = os.path.join(GPR, "rasterize_points.cu")
    o = os.path.join(obj, "rasterize_points.o")
    if _stale(o, [rp, os.path.join(SHIM, "wrap_rasterize_points.cpp")] + shim_deps):
        _run(common + inc + [abi, "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=_C",
                             '-DG2PC_REF_RASTERIZE_POINTS_CU="%s"' % rp, "-c",
                             os.path.join(SHIM, "wrap_rasterize_points.cpp"), "-o", o])
    objs.append(o)
    ext = os.path.join(GPR, "ext.cpp")
    o = os.path.join(obj, "ext.o")
    if _stale(o, [ext]):
        _run(common + inc + [abi, "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=_C", "-x", "c++", "-c", ext, "-o", o])
    objs.append(o)
    so = os.path.join(dst, "_C.so")
    if _stale(so, objs):
        _run(["g++", "-shared", "-Wl,-Bsymbolic", "-o", so] + objs + link)
    return dst


def source_digest() -> str:
    """sha256 over the reference files the module is built from (stored in the goldens, checked where /root/reference exists)."""
    h = hashlib.sha256()
    for rel in ("cuda_rasterizer/forward.cu", "cuda_rasterizer/rasterizer_impl.cu", "cuda_rasterizer/auxiliary.h",
                "cuda_rasterizer/config.h", "rasterize_points.cu", "ext.cpp", "gaussian_pointcloud_rasterization/__init__.py"):
        h.update(open(os.path.join(GPR, rel), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    for v in ("synced", "verbatim"):
        for f in (True, False):
            print(build(v, f))
