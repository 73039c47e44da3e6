"""Multi-GPU shard mode exercised on ONE GPU: every rank linearises a contiguous shard of the factors,
the partial H / b / chi2 are summed by the all-reduce hook, the solve is replicated.
  * the shards' partial systems add up to the unsharded system (g2o graph, mixed GTSAM graph, VIO graph),
  * two ranks as two host threads with a barrier-based hook run the full LM and stay bit-identical to each other and
    within rounding of the unsharded run,
  * two torch.distributed processes (gloo standing in for RCCL on the 1-GPU box) run bench.py --shard end to end."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from tests.test_gpu_parity import synth, make_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partial_systems_add_up_g2o(world):
    g = synth(3000, 5, 4, seed=11)
    H, b, chi = make_gpu(g).read_system()
    Hs = np.zeros_like(H); bs = np.zeros_like(b); cs = 0.0
    for r in range(world):
        gr = make_gpu(g)
        gr.set_shard(r, world)
        h, bb, c = gr.read_system()
        Hs += h; bs += bb; cs += c
    np.testing.assert_allclose(Hs, H, rtol=0, atol=1e-12 * np.abs(H).max())
    np.testing.assert_allclose(bs, b, rtol=0, atol=1e-12 * np.abs(b).max())
    assert abs(cs - chi) <= 1e-13 * chi


def test_partial_systems_add_up_vio():
    from tests.util import vio_graph
    from tests.test_gpu_imu import vio_gpu
    g = vio_graph(np.random.default_rng(1), n_kf=9, with_planes=True)
    H, b, chi = vio_gpu(g).read_system()
    Hs = np.zeros_like(H); bs = np.zeros_like(b); cs = 0.0
    for r in range(3):
        gr = vio_gpu(g)
        gr.set_shard(r, 3)
        h, bb, c = gr.read_system()
        Hs += h; bs += bb; cs += c
    mask = np.abs(H) < 1e13                                   # leave the 1e14 prior entries to a relative check
    np.testing.assert_allclose(Hs[mask], H[mask], rtol=0, atol=1e-9 * np.abs(H[mask]).max())
    np.testing.assert_allclose(Hs[~mask], H[~mask], rtol=1e-12)
    np.testing.assert_allclose(bs, b, rtol=0, atol=1e-9 * max(1.0, np.abs(b).max()))
    assert abs(cs - chi) <= 1e-12 * chi


def test_two_ranks_as_threads_full_lm():
    g = synth(2000, 5, 4, seed=12)
    ref = make_gpu(g)
    ref.optimize(4)
    world = 2
    staging = [None] * world
    barrier = threading.Barrier(world)
    out = [None] * world

    def run(rank):
        gr = make_gpu(g)

        def hook(ptr, n):                                     # all-reduce through host memory, fixed summation order
            t = G.device_tensor(ptr, n)
            staging[rank] = t.cpu()
            barrier.wait()
            total = staging[0] + staging[1]
            barrier.wait()
            t.copy_(total)
            return 0
        gr.set_shard(rank, world, hook)
        rc, st = gr.optimize(4)
        out[rank] = (rc, gr.get_poses().copy(), np.array(gr.trace()[0]))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert out[0] is not None and out[1] is not None
    assert out[0][0] == out[1][0] == 4
    np.testing.assert_array_equal(out[0][1], out[1][1])       # ranks stay bit-identical
    np.testing.assert_allclose(out[0][2], ref.trace()[0], rtol=1e-10)
    np.testing.assert_allclose(out[0][1], ref.get_poses(), atol=1e-8)


def test_bench_shard_two_processes_gloo():
    """bench.py --shard under torch.distributed with 2 ranks sharing the box's single GPU (gloo stands in for RCCL)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--poses", "4000",
           "--shard", "--backend", "gloo", "--cpu-iters", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"].startswith("factor shards")
    assert d["final_chi2"] < d["initial_chi2"]
