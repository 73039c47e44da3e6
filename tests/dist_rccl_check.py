"""One rank of a multi-GPU correctness run of the distributed mode (launched by tests/test_gpu_shard.py under
torch.distributed.run, one process per GPU): the reference's LM schedule on ONE graph distributed over WORLD_SIZE GPUs,
collectives through libfgo's own RCCL communicator enqueued on the context's stream (--transport rccl, the product path on
a multi-GPU node) or through the torch hook (--transport hook: the same script on a single-GPU box, both ranks on GPU 0,
gloo carrying the sums), against the single-GPU run of the same graph on this rank's GPU:
  chi2 trajectory 1e-10 relative, poses 1e-8, final chi2 1e-10, identical scalars / decisions / poses on every rank.
torch.distributed (gloo) only carries the 128-byte RCCL id and the cross-rank comparison of the results.
Solve site being scaled: g2o/g2o_graph.cpp:241-252; VIO variant: gtsam/gtsam_graph.cpp:1784-1788."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import graph_slam_amd as G  # noqa: E402
from tests.test_gpu_parity import synth, make_gpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", default="rccl", choices=["rccl", "hook"])
    ap.add_argument("--poses", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=14)
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--vio-kf", type=int, default=60, help="0: skip the VIO (GTSAM-semantics) graph")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % ndev
    if args.transport == "rccl" and ndev < world:
        raise SystemExit("dist_rccl_check: RCCL needs one GPU per rank (%d visible, world %d)" % (ndev, world))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo")
    def sharded(gr):
        gr.set_shard(rank, world, None if args.transport == "rccl" else G.torch_allreduce_hook(dev))
        if args.transport == "rccl":
            # one id per communicator (the bootstrap thread behind an id serves a single rendezvous): drawn by rank 0, carried by gloo
            uid = [G.dist_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            gr.init_rccl(uid[0])
        return gr

    def same_on_all_ranks(name, a):
        box = [None] * world
        dist.all_gather_object(box, np.asarray(a).tobytes())
        assert all(b == box[0] for b in box), "%s differs between the ranks" % name

    report = {"world": world, "transport": args.transport, "device": dev}
    # ---- g2o path: 5 x optimize(2), the reference's cadence
    g = synth(args.poses, 5, 4, seed=args.seed)
    ref = make_gpu(g, device=dev)
    ref_tr = []
    for _ in range(args.calls):
        rc, _st = ref.optimize(2); assert rc == 2
        ref_tr += list(ref.trace()[0])
    gr = sharded(G.Graph(device=dev))
    gr.add_poses(g["poses"], g["fixed"]); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    tr, trials, xg = [], 0, 0.0
    for _ in range(args.calls):
        rc, st = gr.optimize(2); assert rc == 2
        tr += list(gr.trace()[0]); trials += st.trials; xg += st.reserved[2]
    tr = np.array(tr); poses = gr.get_poses().copy(); chi = gr.chi2()
    same_on_all_ranks("chi2 trajectory", tr); same_on_all_ranks("poses", poses); same_on_all_ranks("trial count", [trials])
    np.testing.assert_allclose(tr, np.array(ref_tr), rtol=1e-10)
    np.testing.assert_allclose(poses, ref.get_poses(), atol=1e-8)
    assert abs(chi - ref.chi2()) <= 1e-10 * ref.chi2()
    assert xg > 0
    report["g2o"] = {"poses": args.poses, "chi2_first": float(tr[0]), "chi2_last": float(tr[-1]), "trials": trials,
                     "max_rel_chi2_diff_vs_1gpu": float(np.max(np.abs(tr - np.array(ref_tr)) / np.array(ref_tr))),
                     "max_pose_diff_vs_1gpu": float(np.abs(poses - ref.get_poses()).max()), "bytes_per_rank_per_trial": xg / max(trials, 1)}
    gr.close(); ref.close()
    # ---- GTSAM path: VIO graph (IMU + between + plane factors, priors), LM with GTSAM's defaults
    if args.vio_kf > 0:
        from tests.util import vio_graph
        from tests.test_gpu_imu import vio_gpu
        gv = vio_graph(np.random.default_rng(5), n_kf=args.vio_kf, with_planes=True)
        ref = vio_gpu(gv, device=dev)
        e0 = ref.error(); ref.optimize_gtsam(20)
        gr = sharded(vio_gpu(gv, device=dev))
        e = gr.error(); gr.optimize_gtsam(20)
        same_on_all_ranks("VIO error", [e, gr.error()]); same_on_all_ranks("VIO values", gr.get_poses())
        assert abs(e - e0) <= 1e-10 * e0
        np.testing.assert_allclose(np.array(gr.trace()[0]), ref.trace()[0], rtol=1e-8)
        assert abs(gr.error() - ref.error()) <= 1e-8 * max(ref.error(), 1.0)
        np.testing.assert_allclose(gr.get_poses(), ref.get_poses(), atol=1e-7)
        report["vio"] = {"keyframes": args.vio_kf, "error_first": e, "error_last": gr.error()}
        gr.close(); ref.close()
    dist.barrier()
    if rank == 0:
        print(json.dumps(report))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
