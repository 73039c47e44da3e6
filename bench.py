#!/usr/bin/env python3
"""Benchmark of the hot path: Levenberg-Marquardt iterations/s on the 100k-pose / 1M-edge SE3 pose graph
(BASELINE.json configs[1]), synthetic "Manhattan-3D" data, f64.

A step = ONE LM iteration of the reference's solver loop (g2o/g2o_graph.cpp:246-249): linearise all edges,
assemble H and b, damp, block-sparse Cholesky, two triangular solves, oplus on every vertex, chi2, rho test.
The graph structure phase (ordering + symbolic factorisation + upload) is done before the timed region
and reported separately (`t_symbolic_s`), as SURVEY.md §8d prescribes; inputs are resident in HBM when the
timed region starts.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def run_iterations(gr, k):
    """exactly k LM iterations (a call may stop early on g2o's 'Terminate'; continue like the reference's loop)"""
    done, last = 0, None
    while done < k:
        rc, st = gr.optimize(k - done)
        done += max(rc, 1)
        last = st
    return last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--poses", type=int, default=100000)
    ap.add_argument("--lookback", type=int, default=5)
    ap.add_argument("--loops", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=3, help="oracle iterations for cpu_baseline (0 = skip)")
    ap.add_argument("--phase-reps", type=int, default=5)
    ap.add_argument("--shard", action="store_true",
                    help="N > 1: ONE graph, factors sharded by pose-block column, Hessian all-reduce (strong scaling) "
                         "instead of the default one-graph-per-GPU replicas (weak scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for 1-GPU smoke tests)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        dev = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")

    import graph_slam_amd as G

    dev = local_rank % torch.cuda.device_count()
    shard = args.shard and world > 1
    # default for N > 1: every rank optimises its own graph of the named size (replicas, weak scaling, no collective).
    # --shard: one graph; each rank linearises a contiguous range of pose-block columns, the partial H / b / chi2 are
    # all-reduced (RCCL) and the solve is replicated (DESIGN.md "Multi-GPU").
    # (replicas all take the SAME graph -- seed 42, the configuration the metric is quoted on -- so that the per-GPU work
    #  is exactly the same for every N: other seeds give graphs whose elimination trees are 31 .. 45 levels deep)
    g = G.synth_manhattan3d(args.poses, args.lookback, args.loops, seed=42)
    n, e = args.poses, len(g["ei"])
    fixed = np.zeros(n, np.uint8); fixed[0] = 1                    # CGraphG2O::firstNode

    def fresh():
        gr = G.Graph(device=dev)
        if shard:
            gr.set_shard(rank, world, G.torch_allreduce_hook(dev))
        gr.add_poses(g["poses"], fixed)
        gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
        return gr

    gr = fresh()
    chi0 = gr.chi2()                                              # builds the structure, uploads to HBM
    sst = gr.stats()
    t_symbolic, t_upload = sst.t_symbolic, sst.t_upload
    if args.warmup > 0:
        run_iterations(gr, args.warmup)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    st = run_iterations(gr, args.steps)
    torch.cuda.synchronize()          # fgo_optimize returns after its own stream sync; belt and braces
    t1 = time.perf_counter()
    sync()
    dt = t1 - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    chi_final = st.chi2_final
    trials = st.trials

    # per-phase device times (HIP events on the library's stream); in shard mode every rank takes part because a
    # phase may need a collective
    ms = {p: gr.bench_phase(p, args.phase_reps) for p in (0, 1, 2)} if (rank == 0 or shard) else None

    out = None
    if rank == 0:
        value = (1 if shard else world) * args.steps / dt
        # ---- roofline of the dominant phase, measured live with HIP events on the library's stream
        # phase 1 is what a trial runs: the level-by-level factor sweep with the forward solve fused in (the right-hand
        # side rides along as one more matrix row); phase 2 is the backward sweep
        names = {0: "k_linearize", 1: "factor sweep + fused forward solve (k_chol_leaf, k_chol_acc, k_panel_tri, k_panel_rows)",
                 2: "backward solve sweep (k_solve_bwd, k_bwd_ext, k_bwd_tri)"}
        bytes_ = {0: sst.bytes_linearize, 1: sst.bytes_factor, 2: sst.bytes_solve}
        dom = max(ms, key=lambda p: ms[p])
        achieved = bytes_[dom] / (ms[dom] * 1e-3) / 1e9
        # HBM bytes of the dominant phase from the PMC counters (FETCH_SIZE / WRITE_SIZE passes, profiles/): a committed
        # measurement of this exact workload; null for any other size
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_cfg2.json")) as fh:
                pm = json.load(fh)
            w = pm["workload"]
            if dom == 1 and (w["poses"], w["lookback"], w["loops"]) == (args.poses, args.lookback, args.loops):
                traffic = pm["hbm_bytes_per_sweep"]
        except (OSError, ValueError, KeyError):
            traffic = None
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_pass": bytes_[dom], "ms_per_pass": ms[dom],
                    "phases_ms": {names[p]: ms[p] for p in ms},
                    "phases_GBs": {names[p]: bytes_[p] / (ms[p] * 1e-3) / 1e9 for p in ms}}
        cpu = None
        chi_rel = None
        if world == 1 and args.cpu_iters > 0:
            # ---- CPU baseline: the oracle (g2o-semantics port, 1 thread) on the SAME graph, bounded sample;
            #      also gives the final-chi2 relative error after the same iteration count from the same start
            from tests import orc_binding as orc
            po = orc.Problem(g["poses"], fixed, g["ei"].astype(np.int32), g["ej"].astype(np.int32), g["meas"], g["info"])
            tc0 = time.perf_counter()
            rc, so = po.optimize(args.cpu_iters)
            tc = time.perf_counter() - tc0
            cpu_it = max(rc, 1)
            cpu = {"value": cpu_it / (tc - so.t_symbolic), "unit": "iterations/s", "cores": 1, "kind": "port",
                   "sample": "%d LM iterations of the same %d-pose / %d-edge graph (oracle: AMD + simplicial "
                             "sparse Cholesky, 1 thread); symbolic %.2fs excluded like on the GPU side"
                             % (cpu_it, n, e, so.t_symbolic),
                   "seconds": tc, "t_symbolic_s": so.t_symbolic, "nnz_L_scalar": so.nnz_L_scalar}
            g2 = fresh()
            r2, s2 = g2.optimize(args.cpu_iters)
            chi_rel = abs(s2.chi2_final - so.chi2_final) / so.chi2_final
            g2.close()
        out = {
            "metric": "Gauss-Newton iterations/s + final chi2 rel-err, 100k-pose SE3 graph",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dk-pose / %.2fM-edge synthetic Manhattan-3D SE3 pose graph, 1xMI355X per replica"
                                   % (n // 1000, e / 1e6),
                       "poses": n, "edges": e, "parallelism": ("factor shards x%d + Hessian all-reduce, replicated solve" % world) if shard else
                                       ("replicas x%d" % world if world > 1 else "single GPU"),
                       "lm_trials_in_timed_region": trials},
            "final_chi2": chi_final, "initial_chi2": chi0, "final_chi2_rel_err_vs_cpu_oracle": chi_rel,
            "t_symbolic_s": t_symbolic, "t_upload_s": t_upload,
            "structure": {"nnz_H_blocks": sst.nnz_H_blocks, "nnz_L_blocks": sst.nnz_L_blocks,
                          "update_ops": sst.n_update_ops, "levels": sst.n_levels, "tasks": sst.n_tasks,
                          "ordering": "nested dissection + leaf minimum degree"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    gr.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
