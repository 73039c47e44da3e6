"""Edge cases of the g2o path against the oracle: the smallest graphs, hub (star) topologies that drive the long-list
accumulate role and dense top panels, disconnected components, a chain (the deepest possible elimination tree), a
complete graph (one dense panel hierarchy)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from tests.test_gpu_parity import make_gpu, make_orc
from tests.util import pose_mul, pose_inv, noisy, random_info, info_ut


def build(rng, truth, pairs, fixed, noise=0.02, init_noise=0.05):
    meas, info = [], []
    for a, b in pairs:
        meas.append(noisy(rng, pose_mul(pose_inv(truth[a]), truth[b]), noise, noise * 0.5))
        info.append(info_ut(random_info(rng)))
    init = np.array([noisy(rng, t, init_noise, init_noise * 0.3) for t in truth])
    fx = np.zeros(len(truth), np.uint8)
    fx[list(fixed)] = 1
    init[list(fixed)] = np.asarray(truth)[list(fixed)]
    return dict(poses=init, fixed=fx, ei=np.array([p[0] for p in pairs], np.int32), ej=np.array([p[1] for p in pairs], np.int32),
                meas=np.array(meas), info=np.array(info))


def random_truth(rng, n, spread=3.0):
    out = []
    for _ in range(n):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = rng.uniform(-1.0, 1.0)
        out.append(np.concatenate([rng.normal(size=3) * spread, ax * np.sin(ang / 2), [np.cos(ang / 2)]]))
    return np.array(out)


def check_against_oracle(g, iters=4, atol=1e-7):
    gr, po = make_gpu(g), make_orc(g)
    c0 = gr.chi2()
    assert abs(c0 - po.chi2()) <= 1e-11 * max(po.chi2(), 1e-30)
    rc, st = gr.optimize(iters)
    ro, so = po.optimize(iters)
    assert rc == ro
    assert abs(st.chi2_final - so.chi2_final) <= 1e-8 * max(so.chi2_final, 1e-12)
    np.testing.assert_allclose(gr.get_poses()[:, :3], po.get_poses()[:, :3], atol=atol)
    return st


def test_two_poses_one_edge():
    rng = np.random.default_rng(1)
    check_against_oracle(build(rng, random_truth(rng, 2), [(0, 1)], fixed=[0]))


def test_three_poses_triangle_second_fixed():
    rng = np.random.default_rng(2)
    check_against_oracle(build(rng, random_truth(rng, 3), [(0, 1), (1, 2), (0, 2)], fixed=[1]))


@pytest.mark.parametrize("n", [200, 3000])
def test_star_hub(n):
    """one pose measured against every other one (+ a sparse ring): the hub column collects thousands of updates"""
    rng = np.random.default_rng(3)
    truth = random_truth(rng, n)
    pairs = [(0, k) for k in range(1, n)] + [(k, k + 1) for k in range(1, n - 1, 3)]
    st = check_against_oracle(build(rng, truth, pairs, fixed=[1]), iters=3)
    assert st.n_free == n - 1


def test_several_hubs_fully_connected_among_themselves():
    rng = np.random.default_rng(4)
    n, hubs = 1500, 12
    truth = random_truth(rng, n)
    pairs = [(a, b) for a in range(hubs) for b in range(a + 1, hubs)]
    pairs += [(int(rng.integers(0, hubs)), k) for k in range(hubs, n)] + [(int(rng.integers(0, hubs)), k) for k in range(hubs, n, 2)]
    pairs = sorted(set(pairs))
    check_against_oracle(build(rng, truth, pairs, fixed=[0]), iters=3)


def test_disconnected_components_each_with_a_fixed_pose():
    rng = np.random.default_rng(5)
    truth = random_truth(rng, 90)
    pairs = [(k, k + 1) for k in range(0, 29)] + [(k, k + 1) for k in range(30, 59)] + [(k, k + 1) for k in range(60, 89)]
    pairs += [(0, 17), (33, 55), (61, 80), (62, 88)]
    check_against_oracle(build(rng, truth, pairs, fixed=[0, 45, 89]))


def test_long_chain_deep_tree():
    """pure odometry chain: nested dissection still has to produce a valid (deep, skinny) schedule"""
    rng = np.random.default_rng(6)
    n = 5000
    truth = [np.array([0, 0, 0, 0, 0, 0, 1.0])]
    for _ in range(1, n):
        truth.append(pose_mul(truth[-1], np.concatenate([[0.5, 0, 0], [0, 0, np.sin(0.02), np.cos(0.02)]])))
    pairs = [(k, k + 1) for k in range(n - 1)]
    check_against_oracle(build(rng, np.array(truth), pairs, fixed=[0], noise=0.01, init_noise=0.02), iters=3, atol=1e-6)


def test_complete_graph_is_one_dense_hierarchy():
    rng = np.random.default_rng(7)
    n = 70
    truth = random_truth(rng, n, spread=1.0)
    pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    st = check_against_oracle(build(rng, truth, pairs, fixed=[0]), iters=3)
    assert st.nnz_L_blocks == (n - 1) * n // 2


def test_all_poses_fixed_is_an_error_and_chi2_still_works():
    rng = np.random.default_rng(8)
    g = build(rng, random_truth(rng, 4), [(0, 1), (1, 2), (2, 3)], fixed=[0, 1, 2, 3])
    gr = make_gpu(g)
    with pytest.raises(G.FgoError):
        gr.optimize(2)
