"""Round-4 entry points of the C ABI that the other test files do not reach: the measured copy bandwidth (bench.py's HBM denominator), the explicit
Retriangulate mode with its argument checks, the timing probe that skips reductions (a no-op without a communicator)."""
import ctypes as C

import numpy as np
import pytest

from spherical_sfm_amd import _lib, ba, synth

pytestmark = pytest.mark.gpu


def test_copy_bandwidth_probe(gpu_ctx):
    g = C.c_double(0.0)
    assert _lib.lib().ssfm_debug_copy_bandwidth(gpu_ctx._p, 256 << 20, 3, C.byref(g)) == 0
    assert 1500.0 < g.value < 8000.0                      # GB/s, read + written bytes: an MI355X copies 4.5-6 TB/s; above the 8 TB/s nominal would be a bug in the probe
    assert _lib.lib().ssfm_debug_copy_bandwidth(gpu_ctx._p, 16, 3, C.byref(g)) != 0       # refuses nonsense sizes


def test_retriangulate_mode_argument(gpu_ctx):
    p = synth.make_circle(60, 600, 6, rot_noise_deg=0.0, pixel_noise=0.3, seed=4)
    Xt, nt = ba.retriangulate(gpu_ctx, p, mode=ba.RETRI_MODE_TRACE)
    Xd, nd = ba.retriangulate(gpu_ctx, p)
    assert np.array_equal(Xt, Xd) and np.array_equal(nt, nd)                 # the default is the trace replay
    Xe, ne = ba.retriangulate(gpu_ctx, p, mode=ba.RETRI_MODE_ENUMERATE)
    assert (np.linalg.norm(Xe - Xt, axis=1) / np.linalg.norm(Xt, axis=1)).max() < 1e-3 and (ne == nt).mean() > 0.99
    with pytest.raises(Exception):
        ba.retriangulate(gpu_ctx, p, mode=7)
    b = ba._ProblemBuffers(p)
    st = np.zeros(2 * len(b.pts), np.uint32)
    rc = _lib.lib().ssfm_retriangulate_mode(gpu_ctx._p, C.byref(b.c), ba.RETRI_MODE_ENUMERATE, None, st.ctypes.data_as(C.POINTER(C.c_uint32)), None)
    assert rc != 0                                        # the enumerating mode has no trace to report


def test_timing_probe_switch_is_harmless_without_a_communicator(gpu_ctx):
    p = synth.make_circle(60, 1500, 6, spherical=False, focal_fixed=True, seed=8)
    c0, p0, f0, s0 = ba.optimize(gpu_ctx, p)
    assert _lib.lib().ssfm_debug_timing_skip_collectives(gpu_ctx._p, 1) == 0
    try:
        c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    finally:
        assert _lib.lib().ssfm_debug_timing_skip_collectives(gpu_ctx._p, 0) == 0
    assert s1["iterations"] == s0["iterations"] and np.abs(c1 - c0).max() <= 1e-9 * np.abs(c0).max()
