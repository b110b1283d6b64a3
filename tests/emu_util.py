"""TEST INFRASTRUCTURE: builds tests/hipemu/libg2pc_emu.so (the product .hip sources compiled by g++
against the fiber emulator) and routes g2pc._native to it for the duration of a test module."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def build_emu(tag: str = "", defs: str = "") -> str:
    """tag / defs: a second emulator library built with other build-time switches (-D...), e.g. ("_r8", "-DG2PC_BK_EMIT_RMAX=8")."""
    if os.environ.get("G2PC_EMU_LIB") and not tag:          # a prebuilt variant, e.g. the AddressSanitizer build of tools/emu_asan.sh
        return os.environ["G2PC_EMU_LIB"]
    env = dict(os.environ, EMU_TAG=tag, EMU_DEFS=defs)
    out = subprocess.run(["bash", os.path.join(HERE, "hipemu", "build_emu.sh")], capture_output=True, text=True, env=env)
    if out.returncode != 0:
        raise RuntimeError("emulator build failed:\n" + out.stdout + out.stderr)
    return os.path.join(HERE, "hipemu", "libg2pc_emu%s.so" % tag)


@pytest.fixture(scope="module")
def emu():
    from g2pc import _native as nv
    saved = (nv._LIB, nv._EMULATED)
    nv._inject_for_tests(build_emu())
    yield nv
    nv._LIB, nv._EMULATED = saved


@pytest.fixture()
def emu_exp():
    """The EXPERIMENTS build of the emulator (-DG2PC_EXPERIMENTS: csrc/experiments/ -- blend kernel variants, 1 / 4 sub-blocks per
    wave, wide radix digits, the process-global knobs) for one test; the library routed before it is restored afterwards."""
    from g2pc import _native as nv
    import gauss_render
    saved = (nv._LIB, nv._EMULATED)
    gauss_render.clear_context_pool()
    nv._inject_for_tests(build_emu("_exp", "-DG2PC_EXPERIMENTS"))
    nv.experiments()
    yield nv
    gauss_render.clear_context_pool()
    nv._LIB, nv._EMULATED = saved
