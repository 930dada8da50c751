"""The hot kernels of the BA iteration carry no scratch and no register spills (read from the built library's code objects: no GPU needed).

Round 6 found 48 B of scratch in every k_schur_gram variant: eight camera ids behind a select chain that the compiler had turned into a per-lane scratch array, i.e. a
chain of scratch / global round trips at the head of every task (profiles/r06_notes.md r06i).  Nothing but the ISA showed it; this test would have."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_resources as KR  # noqa: E402

LIB = os.path.join(ROOT, "spherical_sfm_amd", "libssfm_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(KR.READELF)), reason="needs the built library and llvm-readelf")


@pytest.fixture(scope="module")
def kernels():
    ks = {k["short"]: k for k in KR.kernels(LIB).values()}
    assert len(ks) > 100, "the library's gfx950 code objects were not found"
    return ks


# kernels of one LM iteration on the bench configurations (config 2, focal-free, spherical) and the task kernels of the ragged / large problems
CLEAN = ["k_point_lin<3>", "k_schur_gram<6, 2, 3, 0>", "k_schur_gram<6, 2, 0, 0>", "k_schur_gram<6, 1, 2, 0>", "k_schur_gram<6, 1, 0, 0>",
         "k_schur_gram<3, 1, 0, 0>", "k_schur_gram<3, 1, 2, 0>", "k_schur_gram<3, 2, 0, 0>", "k_schur_gram<3, 2, 3, 0>", "k_schur_gram<3, 3, 0, 0>", "k_schur_gram_any<3>",
         "k_finalize_gather<6, 0, 0>", "k_finalize_gather<6, 0, 1>", "k_finalize_gather<3, 1, 0>", "k_band_chol_v2<6, 2, 0, 3>", "k_band_chol_v2<3, 2, 0, 3>",
         "k_band_back_v2<6, 1>", "k_band_back_v2<3, 1>", "k_arrow_update<6>", "k_arrow_update<3>", "k_point_backsub<6, 1>", "k_point_backsub<6, 2>", "k_gram_backsub2<3>",
         "k_publish", "k_schur_pairs2<6, 2>", "k_cam_sums2<6>", "k_ring_cr_elim<6, 2>", "k_ring_cr_tail<6, 2>", "k_ring_cr_back<6, 2>", "k_sub_spike_fwd<6, 2>",
         "k_sub_sep_assemble_mfma<6, 2>", "k_sub_apply_left<6, 2>"]
# known, bounded: the six-tile Gram kernel sits at its 256 registers (13-14 values spilled around the loop's hand-overs), the 6-dof grouped back substitution spills four
BOUNDED = {"k_schur_gram<6, 3, 0, 0>": 64, "k_schur_gram_any<6>": 64, "k_gram_backsub2<6>": 32}


@pytest.mark.parametrize("name", CLEAN)
def test_hot_kernel_has_no_scratch(kernels, name):
    k = kernels[name]
    assert k["scratch"] == 0 and k["vgpr_spill"] == 0 and not k["dynamic_stack"], k


@pytest.mark.parametrize("name", sorted(BOUNDED))
def test_register_bound_kernels_stay_bounded(kernels, name):
    k = kernels[name]
    assert k["scratch"] <= BOUNDED[name] and not k["dynamic_stack"], k


def test_register_budgets(kernels):
    # two waves per SIMD for the Gram tasks (<= 256 registers), four for the grouped back substitution (<= 128)
    for n, k in kernels.items():
        if n.startswith("k_schur_gram"): assert k["vgpr"] + k["agpr"] <= 256, (n, k)
        if n.startswith("k_gram_backsub2"): assert k["vgpr"] + k["agpr"] <= 128, (n, k)
