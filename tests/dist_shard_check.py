"""Run under torch.distributed (gloo, CPU): the factor-shard decomposition used by the multi-GPU mode, checked with a
real collective.  Each rank evaluates the oracle's dense system on ITS contiguous edge shard (fgo_shard_range, the same
helper libfgo uses), the partial H / b / chi2 are all-reduced, and every rank compares with the unsharded system."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import graph_slam_amd as G          # noqa: E402  (host-only entry points: generator + shard helper)
from tests import orc_binding as orc  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 80
    g = G.synth_manhattan3d(n, 4, 2, seed=5)
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    ei, ej = g["ei"].astype(np.int32), g["ej"].astype(np.int32)
    lo, hi = C.c_int64(), C.c_int64()
    assert G.lib.fgo_shard_range(len(ei), rank, world, C.byref(lo), C.byref(hi)) == 0
    s = slice(lo.value, hi.value)
    part = orc.Problem(g["poses"], fixed, ei[s], ej[s], g["meas"][s], g["info"][s])
    H, b = part.dense_system()
    chi = part.chi2()
    tH, tb, tc = torch.from_numpy(H), torch.from_numpy(b), torch.tensor([chi], dtype=torch.float64)
    for t in (tH, tb, tc):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    full = orc.Problem(g["poses"], fixed, ei, ej, g["meas"], g["info"])
    Hf, bf = full.dense_system()
    ok = (np.abs(tH.numpy() - Hf).max() <= 1e-12 * np.abs(Hf).max() and np.abs(tb.numpy() - bf).max() <= 1e-12 * np.abs(bf).max()
          and abs(tc.item() - full.chi2()) <= 1e-12 * full.chi2())
    # every rank must see the same reduced bits (what keeps the replicated solves in lockstep)
    digest = torch.tensor([float(np.float64(tH.numpy().sum()))], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    ok = ok and all(float(x) == float(gathered[0]) for x in gathered)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
