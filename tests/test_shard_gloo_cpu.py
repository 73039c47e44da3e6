"""world_size-2 and -3 gloo runs on CPU of the multi-GPU mode's algorithm -- libfgo's own domain decomposition, numpy
elimination per rank, a real all-reduce of the separator system (tests/dist_shard_check.py) -- plus host-only checks of
the decomposition."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29611), (3, 29612)])
def test_factor_shards_allreduce_gloo(world, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_shard_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_decomposition_separates_domains(world):
    """no edge joins two different domains; every rank gets work; the top is a small fraction of the graph"""
    import numpy as np
    import graph_slam_amd as G
    n = 20000
    g = G.synth_manhattan3d(n, 5, 4, seed=7)
    grp = G.debug_partition(n, g["ei"], g["ej"], world)
    a, b = grp[g["ei"]], grp[g["ej"]]
    cross = (a != b) & (a < world) & (b < world)
    assert not cross.any()
    counts = np.bincount(grp, minlength=world + 1)
    assert (counts[:world] > 0).all()
    assert counts[world] < 0.15 * n, counts
    assert counts[:world].max() < 2.5 * counts[:world].mean(), counts
