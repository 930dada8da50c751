"""Host code of the library under sanitizers (CPU build only; GPU sanitizers are not available on the pool): the planner
(ba_flatten.h: flatten rules, sharding, S structure, pair lists, wave-task tables, std::thread sections) and the track builder,
on random problems with out-of-range ids, duplicate keys, zero points, unsorted input, 1-3 ranks."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "native", "plan_sanitize.cpp"), os.path.join(ROOT, "spherical_sfm_amd", "csrc", "tracks.cpp")]


@pytest.mark.parametrize("flags,env", [("-fsanitize=address,undefined -fno-sanitize-recover=all", {}), ("-fsanitize=thread", {"SSFM_PLAN_THREADS": "4"})])
def test_planner_and_tracks_under_sanitizers(tmp_path, flags, env):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "plan_sanitize")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", *flags.split(), "-I", os.path.join(ROOT, "include"), *SRC, "-o", exe, "-pthread"],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert run.returncode == 0 and "SANITIZE_OK" in run.stdout, (run.stdout + run.stderr)[-3000:]
    assert "runtime error" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr


def test_mirror_containers_under_sanitizers(tmp_path):
    """IndexedMap / FlatMap (csrc/shim/sfm.h) -- the dense point table and the sorted-vector observation rows behind the std::map interface of the SfM mirror --
    against std::map on random operation sequences, ASan + UBSan."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "shim_maps_sanitize")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", os.path.join(ROOT, "tests", "native", "shim_maps_sanitize.cpp"), "-o", exe],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "SHIM_MAPS_OK" in run.stdout, (run.stdout + run.stderr)[-3000:]
    assert "runtime error" not in run.stderr


@pytest.mark.skipif(not os.path.exists("/root/reference/include/sphericalsfm/sparse.hpp"), reason="the reference tree is only in the build container")
def test_mirror_containers_against_the_references_own_sparse_hpp(tmp_path):
    """VERDICT r5 #7b: the same harness with the REFERENCE's SparseVector / SparseMatrix (include/sphericalsfm/sparse.hpp: std-only, compiled as it stands from
    /root/reference, nothing copied) as the yardstick for the point table and the observation table, on the operations src/sfm.cpp uses.  ASan + UBSan."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "shim_maps_vs_reference")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DSSFM_REF_SPARSE", "-I/root/reference/include",
                         os.path.join(ROOT, "tests", "native", "shim_maps_sanitize.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "SHIM_VS_REFERENCE_SPARSE_OK" in run.stdout and "SHIM_MAPS_OK" in run.stdout, (run.stdout + run.stderr)[-3000:]
    assert "runtime error" not in run.stderr


def test_ring_schedule_against_a_dense_solve_under_sanitizers(tmp_path):
    """Cyclic-reduction schedule of the ring-native reduced solve (csrc/ring_schedule.h; kernels: band_ring.h): the records are replayed with dense loops exactly as the
    kernels read them -- gathered pending updates, couplings from Z or as products of stored F blocks, back substitution in reverse step order -- on random SPD cyclic
    block systems (2 .. 64 separators, several rings at once) and compared with a dense Cholesky solve (<= 1e-10), ASan + UBSan."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "ring_schedule_check")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", os.path.join(ROOT, "tests", "native", "ring_schedule_check.cpp"), "-o", exe],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "RING_SCHEDULE_OK" in run.stdout, (run.stdout + run.stderr)[-3000:]
    assert "runtime error" not in run.stderr


def test_fixed_point_splits_of_the_deterministic_accumulation(tmp_path):
    """csrc/det_acc.h (SSFM_DETERMINISTIC=1): the device adds every addend as fixed-point limbs with integer atomics.  The splits are host-compilable: exact over the
    whole exponent range (long accumulator: 2^-180 .. 2^100, matrix accumulators: to 2^-74 absolute below 2^40), coefficient bounds, limb sums independent of the
    order of the addends, decoded sums accurate to a double's last place, non-finite and out-of-range values refused (the device poisons the sum: NaN).  UBSan on."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "det_acc_check")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=all", os.path.join(ROOT, "tests", "native", "det_acc_check.cpp"), "-o", exe],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "DET_ACC_OK" in run.stdout, (run.stdout + run.stderr)[-3000:]
    assert "runtime error" not in run.stderr
