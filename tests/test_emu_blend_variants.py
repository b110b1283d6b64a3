"""Two-wave form of the dual-list blend (k_blend_py_2w) against the single-wave kernel and the reference's golden vectors, on
the CPU emulator (real barriers between the two waves of a block)."""
from emu_util import emu  # noqa: F401


def test_two_wave_blend_equals_single_wave_blend(emu, golden_dir):
    from blend_variant_checks import run_variants, assert_variants_agree
    assert_variants_agree(run_variants("cpu", golden_dir))
