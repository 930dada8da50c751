"""Worker for tests/test_multirank_gpu.py: one rank of a point-sharded bundle adjustment.
mode 'host': N ranks share GPU 0, reductions go through the host all-reduce hook (gloo).
mode 'rccl1': one rank with a forced 1-rank RCCL communicator (SSFM_COMM_SINGLE_RANK=1) - exercises ncclAllReduce on the solver stream.
mode 'rccl': N ranks on N GPUs with a real N-rank RCCL communicator (needs >= N visible GPUs: the tests skip otherwise)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    mode, out, spherical, focal_fixed = sys.argv[1], sys.argv[2], sys.argv[3] == "1", sys.argv[4] == "1"
    import torch
    from spherical_sfm_amd import ba, synth
    if len(sys.argv) > 5 and sys.argv[5] == "ring":           # one ring of 1000 cameras: the ring-native reduced solve (band_ring.h), replicated on every rank
        prob = synth.make_circle(1000, 20000, 6, spherical=spherical, focal_fixed=focal_fixed, seed=21)
    elif len(sys.argv) > 5 and sys.argv[5] == "weak2":        # the shape bench.py --gpus 2 runs: 2 x config 2's cameras, 8 rings of 75
        prob = synth.make_circle(600, 24000, 6, spherical=spherical, focal_fixed=focal_fixed, seed=21)
    else:
        prob = synth.make_circle(60, 6000, 6, spherical=spherical, focal_fixed=focal_fixed, seed=21)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    ctx = ba.Context(rank if mode == "rccl" else 0)         # 'rccl': one GPU per rank, the real multi-GPU layout
    if mode == "host":
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        def hook(arr, op):
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        ctx.comm_init_host(world, rank, hook)
    elif mode == "rccl1":
        ctx.comm_init(ba.Context.unique_id(), 1, 0)
    elif mode == "rccl":
        # N ranks on N GPUs, reductions by ncclAllReduce over xGMI: the id travels through gloo like bench.py's does through torch.distributed
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [ba.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(box[0], world, rank)
    if len(sys.argv) > 5 and sys.argv[5] == "ransac":
        # ssfm_ransac_batch_sharded: every rank passes the same ragged pair list (one pair below the minimal sample size)
        from spherical_sfm_amd import ransac
        pairs = [synth.make_relative_pose_problem(n, seed=100 + i, noise=1 / 600, outlier_frac=0.3, rotation_deg=10)[:2]
                 for i, n in enumerate([120, 75, 33, 2, 200, 64, 97])]
        o = ransac.estimate_pairs(ctx, pairs, (2 / 600) ** 2, sharded=True, min_num_inliers=12, num_hypotheses=256, mode=int(os.environ.get("RANSAC_MODE", "1")))
        np.savez(out + f".{rank}.npz", E=o["E"], R=o["R"], num_inliers=o["num_inliers"], scores=o["scores"], mask=np.concatenate(o["inliers"]),
                 iterations=o["iterations"], lo_runs=o["lo_runs"])
        if mode in ("host", "rccl"):
            dist.barrier(); dist.destroy_process_group()
        return
    if len(sys.argv) > 5 and sys.argv[5] == "ransac_indexed":
        # ssfm_ransac_batch_indexed_sharded: per-frame feature rays + per-pair match lists, pairs round robin over the ranks
        from spherical_sfm_amd import ransac
        import _pairwise_frames
        a = _pairwise_frames.indexed_problem()
        o = ransac.estimate_indexed(ctx, *a, (2 / 600) ** 2, sharded=True, min_num_inliers=12)
        np.savez(out + f".{rank}.npz", E=o["E"], R=o["R"], num_inliers=o["num_inliers"], scores=o["scores"], mask=o["mask"], iterations=o["iterations"], lo_runs=o["lo_runs"])
        if mode in ("host", "rccl"):
            dist.barrier(); dist.destroy_process_group()
        return
    cams, pts, focal, summ = ba.optimize(ctx, prob)
    np.savez(out + f".{rank}.npz", cams=cams, pts=pts, focal=focal, iterations=summ["iterations"], final_cost=summ["final_cost"],
             initial_cost=summ["initial_cost"], termination=summ["termination"])
    if mode in ("host", "rccl"):
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
