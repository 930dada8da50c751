import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # test-only helper modules (tests/_*.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored); build it when a fresh checkout runs the tests directly
    built = [os.path.join(ROOT, "spherical_sfm_amd", f) for f in ("libssfm_hip.so", "demo_circle", "demo_formats", "demo_focal", "demo_estimator", "run_spherical_sfm", "run_spherical_sfm_uncalib")]
    if not all(os.path.exists(f) for f in built) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "spherical_sfm_amd", "csrc"), "all"])


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand from oracle/*.cpp."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch  # noqa: F401  (share torch's HIP runtime; also gives device queries)
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from spherical_sfm_amd import ba
    ctx = ba.Context(0)
    yield ctx
    ctx.close()
