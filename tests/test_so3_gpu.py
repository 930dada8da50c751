"""Row a9 on the device: so3exp / so3ln (src/so3.cpp:16-69) and Ceres' AngleAxisToRotationMatrix / RotationMatrixToAngleAxis as the HIP
kernels evaluate them (csrc/ssfm_math.h), through ssfm_so3_probe -- against the committed golden vectors, the oracle, and scipy.
Covers theta = 0, 1e-12, 1e-9 (identity branch of so3exp, first-order branch of Ceres), the asin and acos branches of so3ln and its
symmetric-part branch near pi with each of the three "largest diagonal" sub-branches (src/so3.cpp:44-66)."""
import os
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_so3_golden_on_the_device(gpu_ctx):
    from spherical_sfm_amd import ransac
    g = np.load(os.path.join(GOLD, "so3.npz"))
    assert np.abs(ransac.so3_probe(gpu_ctx, "exp", g["r"]) - g["so3exp"]).max() <= 1e-15
    assert np.abs(ransac.so3_probe(gpu_ctx, "ln", g["so3exp"]) - g["so3ln"]).max() <= 1e-9       # theta = pi - 1e-6: asin near 0 amplifies rounding to ~1e-10
    assert np.abs(ransac.so3_probe(gpu_ctx, "aa2R", g["r"]) - g["ceres_R"]).max() <= 1e-15
    assert np.abs(ransac.so3_probe(gpu_ctx, "R2aa", g["so3exp"]) - g["ceres_aa"]).max() <= 1e-12


def test_so3ln_near_pi_all_sub_branches(gpu_ctx, oracle):
    from spherical_sfm_amd import ransac
    rs, branch = [], []
    rng = np.random.default_rng(5)
    for dom in range(3):                          # axis dominated by x, y, z -> the d0 / d1 / d2 sub-branch of src/so3.cpp:44-66
        for theta in (np.pi - 1e-6, np.pi - 1e-3, 3.0, 2.5, np.pi - 1e-9):
            for sgn in (1.0, -1.0):
                ax = 0.3 * rng.normal(size=3); ax[dom] = sgn * 2.0; ax /= np.linalg.norm(ax)
                rs.append(ax * theta); branch.append(dom)
    rs = np.array(rs)
    R = Rotation.from_rotvec(rs).as_matrix()
    cos_angle = (np.trace(R, axis1=1, axis2=2) - 1) / 2
    assert (cos_angle <= -np.sqrt(0.5)).all()     # every case is in the third branch
    diag = np.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]], axis=1) - cos_angle[:, None]
    assert (np.argmax(np.abs(diag), axis=1) == np.array(branch)).all()
    got = ransac.so3_probe(gpu_ctx, "ln", R)
    ref = np.array([oracle.so3ln(Ri) for Ri in R])
    assert np.abs(got - ref).max() <= 1e-9
    # and against scipy: the log of the same rotation (theta -> pi loses digits in sin: 1e-6 rad at pi - 1e-9 is the formula's own accuracy)
    err = [np.linalg.norm(Rotation.from_matrix(Rotation.from_rotvec(a).as_matrix() @ Ri.T).as_rotvec()) for a, Ri in zip(got, R)]
    assert max(err) <= 2e-7
    # Ceres' conversion on the same matrices (quaternion route, every trace branch)
    aa = ransac.so3_probe(gpu_ctx, "R2aa", R)
    assert np.abs(aa - np.array([oracle.rotation_matrix_to_angle_axis(Ri) for Ri in R])).max() <= 1e-12


def test_so3_random_against_scipy_and_oracle(gpu_ctx, oracle):
    from spherical_sfm_amd import ransac
    rng = np.random.default_rng(9)
    rs = rng.normal(size=(2000, 3)); rs *= (rng.uniform(0, np.pi - 1e-3, 2000) / np.linalg.norm(rs, axis=1))[:, None]
    rs[:4] = [[0, 0, 0], [1e-12, 0, 0], [0, 1e-9, 0], [3e-9, -2e-9, 1e-9]]
    R = ransac.so3_probe(gpu_ctx, "exp", rs)
    assert np.abs(R - Rotation.from_rotvec(rs).as_matrix()).max() <= 2e-10       # src/so3.cpp:18 returns I below 1e-10
    assert np.abs(R[:50] - np.array([oracle.so3exp(r) for r in rs[:50]])).max() <= 1e-15
    assert np.abs(ransac.so3_probe(gpu_ctx, "ln", R) - rs).max() <= 1e-9
    Rc = ransac.so3_probe(gpu_ctx, "aa2R", rs)
    assert np.abs(Rc - Rotation.from_rotvec(rs).as_matrix()).max() <= 1e-12
    assert np.abs(ransac.so3_probe(gpu_ctx, "R2aa", Rc) - rs).max() <= 1e-9
