"""Run under torch.distributed (gloo, CPU): the multi-GPU mode's algorithm with a real collective, on the host.
libfgo's decomposition (fgo_debug_partition: the same ordering + symbolic code fgo_set_shard uses, host only) assigns
every free pose to a rank's domain or to the top.  Each rank then does, in numpy, what its GPU does:
  linearise ITS factors (oracle, dense) -> eliminate ITS domain columns (Schur complement onto the top) -> all-reduce of
  the top system (the contributions that cross from a domain into the separator columns) -> solve the top redundantly ->
  back-substitute its own domain -> gather.
The result must equal the dense solve of the whole system on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graph_slam_amd as G          # noqa: E402  (host-only entry points: generator + decomposition)
from tests import orc_binding as orc  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 400
    g = G.synth_manhattan3d(n, 4, 2, seed=5)
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    ei, ej = g["ei"].astype(np.int32), g["ej"].astype(np.int32)
    # free poses 1..n-1 -> hessian index v-1
    keep = (ei != 0) & (ej != 0)
    group = G.debug_partition(n - 1, ei[keep] - 1, ej[keep] - 1, world)
    assert set(np.unique(group)) <= set(range(world + 1))
    pg = np.concatenate([[-1], group])                     # per pose; the fixed pose belongs to nobody
    # a factor belongs to the rank of its domain endpoint; factors among top / fixed poses only are dealt round-robin
    owner = np.full(len(ei), -1)
    for e in range(len(ei)):
        gs = [x for x in (pg[ei[e]], pg[ej[e]]) if 0 <= x < world]
        assert len(set(gs)) <= 1, "a factor spans two domains"
        owner[e] = gs[0] if gs else e % world
    mine = owner == rank
    part = orc.Problem(g["poses"], fixed, ei[mine], ej[mine], g["meas"][mine], g["info"][mine])
    H, b = part.dense_system()
    chi = part.chi2()
    lam = 1e-3
    idx = lambda vs: np.concatenate([np.arange(6 * v, 6 * v + 6) for v in vs]) if len(vs) else np.zeros(0, int)
    dom = idx(np.where(group == rank)[0]); top = idx(np.where(group == world)[0])
    Hdd = H[np.ix_(dom, dom)] + lam * np.eye(len(dom)); Htd = H[np.ix_(top, dom)]
    Htt = H[np.ix_(top, top)] + (lam * np.eye(len(top)) if rank == 0 else 0)
    sol = np.linalg.solve(Hdd, np.column_stack([Htd.T, b[dom]])) if len(dom) else np.zeros((0, len(top) + 1))
    S = torch.from_numpy(Htt - Htd @ sol[:, :-1]); rhs = torch.from_numpy(b[top] - Htd @ sol[:, -1])
    tc = torch.tensor([chi], dtype=torch.float64)
    for t in (S, rhs, tc):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)           # the collective of the multi-GPU mode
    xt = np.linalg.solve(S.numpy(), rhs.numpy())
    xd = sol[:, -1] - sol[:, :-1] @ xt
    x = torch.zeros(6 * (n - 1), dtype=torch.float64)
    x[dom] = torch.from_numpy(xd)
    if rank == 0:
        x[top] = torch.from_numpy(xt)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)               # gather
    full = orc.Problem(g["poses"], fixed, ei, ej, g["meas"], g["info"])
    Hf, bf = full.dense_system()
    ref = np.linalg.solve(Hf + lam * np.eye(len(bf)), bf)
    ok = np.abs(x.numpy() - ref).max() <= 1e-9 * np.abs(ref).max() and abs(tc.item() - full.chi2()) <= 1e-12 * full.chi2()
    ok = ok and (group == rank).sum() > 0
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
