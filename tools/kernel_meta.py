"""Kernel resource usage straight from the shipped library: pulls the gfx950 code objects out of libg2pc.so's clang offload
bundles and reads the AMDGPU metadata notes (llvm-readelf --notes): VGPRs, AGPRs, SGPRs, spills, static LDS per kernel,
and the waves per SIMD those allow (MI355X_MICROARCH.md: 512 VGPRs per SIMD lane, allocation granularity 8, at most 8
waves per SIMD; 160 KB of LDS per CU).

    python tools/kernel_meta.py [substring ...]        # e.g. k_blend_py_dl  ->  JSON, one record per matching kernel
bench.py imports `kernel_meta(name)` for the `vgpr` / `max_waves_per_simd` fields of its roofline block."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "3dgs-to-pc_amd", "g2pc", "libg2pc.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path=LIB, arch="gfx950"):
    """Yields the device ELF images of `arch` found in the library's offload bundles."""
    blob = open(path, "rb").read()
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def all_kernels(path=LIB):
    out = {}
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            block = ".agpr_count:" + block
            rec = {k: int(v) for k, v in re.findall(r"\.(agpr_count|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|"
                                                    r"group_segment_fixed_size|private_segment_fixed_size|max_flat_workgroup_size):\s+(\d+)", block)}
            m = re.search(r"\.name:\s+(\S+)", block)
            if m:
                out[m.group(1)] = rec
    return out


def demangle(names):
    """mangled -> demangled (llvm-cxxfilt where the image has it, else binutils' c++filt; the mangled name itself if neither)."""
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            p = subprocess.run([tool] + list(names), capture_output=True, text=True)
            d = p.stdout.split("\n")[:len(names)]
            if p.returncode == 0 and len(d) == len(names):
                return dict(zip(names, d))
        except Exception:
            pass
    return {n: n for n in names}


def _matches(substring, mangled, demangled):
    """`k_blend_py_dl<4>` names ONE template instance: matched in the demangled name, or -- no demangler -- as the Itanium
    encoding of an integer template argument (k_blend_py_dlILi4E)."""
    if substring in demangled:
        return True
    m = re.fullmatch(r"(.*?)<(\d+)>", substring.split("::")[-1])
    return bool(m) and ("%sILi%sE" % (m.group(1), m.group(2))) in mangled


def occupancy(rec, block_threads=None):
    regs = rec.get("vgpr_count", 0) + rec.get("agpr_count", 0)
    alloc = max(8, (regs + 7) // 8 * 8)
    waves = max(1, min(8, 512 // alloc))
    lds = rec.get("group_segment_fixed_size", 0)
    out = {"waves_per_simd_by_vgpr": waves}
    if lds and block_threads:
        blocks = (160 * 1024) // lds
        out["waves_per_simd_by_lds"] = min(8, blocks * (block_threads // 64) // 4)
        out["max_waves_per_simd"] = min(waves, out["waves_per_simd_by_lds"])
    else:
        out["max_waves_per_simd"] = waves
    return out


def kernel_meta(substring, path=LIB):
    """First kernel whose demangled name contains `substring`: {name, vgpr_count, ..., max_waves_per_simd} or None.  Name a
    template instance with its arguments ("k_blend_py_dl<4>"): a bare "k_blend_py_dl" returns whichever instance comes first."""
    ks = all_kernels(path)
    dm = demangle(list(ks))
    for mangled, rec in sorted(ks.items(), key=lambda kv: dm[kv[0]]):
        if _matches(substring, mangled, dm[mangled]):
            r = dict(rec, name=dm[mangled].split("(")[0])
            r.update(occupancy(rec, rec.get("max_flat_workgroup_size")))
            return r
    return None


if __name__ == "__main__":
    ks = all_kernels()
    dm = demangle(list(ks))
    want = sys.argv[1:]
    for mangled, rec in sorted(ks.items(), key=lambda kv: dm[kv[0]]):
        if not want or any(_matches(w, mangled, dm[mangled]) for w in want):
            r = dict(rec, name=dm[mangled].split("(")[0])
            r.update(occupancy(rec, rec.get("max_flat_workgroup_size")))
            print(json.dumps(r))
