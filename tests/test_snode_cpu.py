"""Host-only checks of the supernodal solver's PLAN (csrc/snode.h: snode_plan through ssfm_snode_plan_probe) and of the algorithm it drives: a numpy restatement of the
kernel's elimination (tests/_snode_ref.py) that walks the plan's own tables reproduces numpy's dense solve on every structure the GPU suite uses."""
import numpy as np
import pytest

import _snode_ref as R
from test_snode_gpu import CASES, window_system


@pytest.mark.parametrize("case", range(len(CASES)))
def test_plan_tables_drive_a_correct_elimination(case):
    from spherical_sfm_amd import ba
    dc, comps, shuffle = CASES[case]
    rp, ci, blk, A, rhs2 = window_system(comps, dc, seed=100 + case, shuffle_ids=shuffle)
    plan = ba.snode_plan_probe(dc, rp, ci)
    assert plan is not None
    nrings = sum(1 for c in comps if c[2]); assert (plan["qtm"] > 0) == (nrings > 0)
    # every camera sits in exactly one node
    cams = plan["node_cam"][plan["node_cam"] >= 0]
    assert sorted(cams.tolist()) == list(range(len(rp) - 1))
    Y = R.solve(plan, blk, rhs2, dc)
    xr = np.linalg.solve(A, rhs2.T).T
    assert np.abs(Y - xr).max() <= 1e-11 * np.abs(xr).max()


def test_config2_plan_shape():
    """four rings of 75 cameras, six-frame tracks: eight workgroups of seven pivots, T of five cameras"""
    from spherical_sfm_amd import ba
    rp, ci, blk, A, rhs2 = window_system([(75, 5, True)] * 4, 6, seed=1)
    plan = ba.snode_plan_probe(6, rp, ci)
    assert plan["nhalf"] == 8 and plan["qtm"] == 30 and plan["S"] == 5
    assert sorted(plan["half_rec"][:, 0].tolist()) == [6, 6, 6, 6, 7, 7, 7, 7]
    assert (plan["half_rec"][:, 2] >= 0).all()


@pytest.mark.parametrize("dc,comps", [(6, [(75, 7, True)]), (6, [(19, 5, True)]), (3, [(80, 11, True)]), (6, [(400, 5, True)])])
def test_plan_refuses(dc, comps):
    from spherical_sfm_amd import ba
    rp, ci, blk, A, rhs2 = window_system(comps, dc, seed=7)
    assert ba.snode_plan_probe(dc, rp, ci) is None


def test_plan_needs_every_workgroup_resident():
    from spherical_sfm_amd import ba
    rp, ci, blk, A, rhs2 = window_system([(40, 5, True)] * 5, 6, seed=2)
    assert ba.snode_plan_probe(6, rp, ci, num_cus=256)["nhalf"] == 10
    assert ba.snode_plan_probe(6, rp, ci, num_cus=8) is None
