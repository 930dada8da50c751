"""world_size-2 test of the multi-GPU design on CPU (gloo): the reduced camera system is a plain sum over point
shards, so all-reducing the per-rank partial systems must reproduce the single-rank system.  Shards come from the
product's own planner (ssfm_ba_plan); the partial systems from the oracle.  This is exactly the exchange the HIP
path does with RCCL (ba_solver.hip: one all-reduce of [S | rhs | ...] per LM iteration)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spherical_sfm_amd import ba, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close(); return port


def _shard(p, used):
    q = p.copy(); m = used.astype(bool)
    q.obs_xy, q.obs_cam, q.obs_pt = p.obs_xy[m], p.obs_cam[m], p.obs_pt[m]
    return q


def _worker(rank, world, port, spherical, focal_fixed, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    p = synth.make_circle(60, 900, 6, spherical=spherical, focal_fixed=focal_fixed)
    info, ids, used, pos = ba.plan(p, world, rank)
    S, rhs = O.ba_reduced_system(_shard(p, used), mu=0.5)
    cost = torch.tensor([O.ba_evaluate(_shard(p, used))[0]], dtype=torch.float64)
    tS, tr = torch.from_numpy(S), torch.from_numpy(rhs)
    dist.all_reduce(tS); dist.all_reduce(tr); dist.all_reduce(cost)
    if rank == 0:
        Sf, rf = O.ba_reduced_system(p, mu=0.5)
        cf = O.ba_evaluate(p)[0]
        out["err_S"] = float(np.abs(tS.numpy() - Sf).max() / np.abs(Sf).max())
        out["err_r"] = float(np.abs(tr.numpy() - rf).max() / np.abs(rf).max())
        out["err_c"] = float(abs(cost.item() - cf) / cf)
        out["nobs"] = int(info["num_observations_used"])
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("spherical,focal_fixed", [(True, False), (False, True)])
def test_partial_reduced_systems_allreduce_to_the_full_one(spherical, focal_fixed):
    port = _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(2, port, spherical, focal_fixed, out), nprocs=2, join=True)
    assert out["err_S"] < 1e-12 and out["err_r"] < 1e-12 and out["err_c"] < 1e-13
    assert out["nobs"] == 2700
