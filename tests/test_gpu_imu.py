"""GPU parity tests of the CombinedImuFactor kernel and VIO-type graphs (poses + velocities + biases + IMU factors +
between factors + plane landmarks + priors: what test_vro_imu_graph.cpp / test_ba_imu_graph.cpp assemble) through the
C-ABI against the oracle.  Product and oracle are handed the SAME 15x15 information matrix (fgo_preint_information: the
Cholesky inverse of preintMeasCov the product uses), so the factor arithmetic is held to 1e-10 like every other factor
type.  One test keeps the independent route (oracle weighted with numpy.linalg.inv of the covariance): the covariance's
condition number (~1e9) bounds the agreement of the two inverses to ~1e-7, hence 1e-6 there."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from tests import orc_binding as orc
from tests.util import vio_graph, mixed_oracle


def vio_gpu(g, **kw):
    gr = G.Graph(**kw)
    K, npl = g["n_kf"], g["n_planes"]
    gr.add_poses(g["values"][:K])
    for k in range(K):                         # insertion order = id order, so dense read-backs line up with the oracle
        gr.add_vec3(K + k, g["values"][K + k, :3])
    for k in range(K):
        gr.add_bias(2 * K + k, g["values"][2 * K + k, :6])
    for p in range(npl):
        gr.add_plane(3 * K + p, g["values"][3 * K + p, :4])
    for k in range(len(g["ei"])):
        i, j, kind = int(g["ei"][k]), int(g["ej"][k]), g["kind"][k]
        if kind == orc.FK_BETWEEN:
            gr.add_edges([i], [j], g["meas"][k:k + 1], g["info"][k:k + 1], tangent_order=G.FGO_TANGENT_GTSAM)
        else:
            gr.add_plane_factor(i, j, g["meas"][k, :4], [1e-4, 0, 0, 1e-4, 0, 1e-4])
    gr.add_prior(0, g["prior_mean"][0], g["prior_info"][0])
    gr.add_prior_vec3(K, g["prior_mean"][1][:3], 1e-3)          # gtsam_graph.cpp:359-364
    gr.add_prior_bias(2 * K, g["prior_mean"][2][:6], 1e-3)
    for f, ids in enumerate(g["imu_ids"]):
        gr.add_imu(ids, g["imu_pre"][f].buf)
    return gr


def same_information(g):
    """hand the oracle the information matrices the product derives from the payloads (fgo_preint_information)"""
    g["imu_info"] = np.array([G.preint_information(p.buf) for p in g["imu_pre"]])
    return g


@pytest.mark.parametrize("seed,planes", [(0, False), (1, True)])
def test_vio_linearization_matches_oracle(seed, planes):
    """same information matrix on both sides: chi2 1e-12, H / b 1e-10 of the largest entry"""
    rng = np.random.default_rng(seed)
    g = same_information(vio_graph(rng, n_kf=7, with_planes=planes))
    gr, po = vio_gpu(g), mixed_oracle(g)
    chi, H, b = gr.linearize()
    Ho, bo = po.dense_system()
    assert abs(chi - po.chi2()) <= 1e-11 * po.chi2()
    mask = np.ones(len(bo), bool); mask[:6] = False               # (pose 0 carries the 1e14 prior)
    sub = np.ix_(mask, mask)
    np.testing.assert_allclose(H[sub], Ho[sub], rtol=0, atol=1e-10 * np.abs(Ho[sub]).max())
    np.testing.assert_allclose(b[mask], bo[mask], rtol=0, atol=1e-10 * np.abs(bo[mask]).max())
    assert abs(gr.chi2() - po.chi2()) <= 1e-11 * po.chi2()       # stand-alone chi2 kernel (IMU part included)


def test_imu_factors_sharing_one_bias_pair():
    """Colouring of the IMU factors (fgo_structure.cpp imu_colour_add): every factor adds its blocks straight into H in the
    launch of its colour.  69 factors that all use the SAME two bias variables (a constant-bias model) conflict pairwise: the
    64 colours run out and the rest take a launch each.  H / b / chi2 against the oracle as above."""
    rng = np.random.default_rng(3)
    g = same_information(vio_graph(rng, n_kf=70, with_planes=False))
    K = g["n_kf"]
    g["imu_ids"][:, 4] = 2 * K
    g["imu_ids"][:, 5] = 2 * K + 1
    assert len(g["imu_ids"]) == K - 1 > 64
    gr, po = vio_gpu(g), mixed_oracle(g)
    chi, H, b = gr.linearize()
    Ho, bo = po.dense_system()
    assert abs(chi - po.chi2()) <= 1e-11 * po.chi2()
    mask = np.ones(len(bo), bool); mask[:6] = False
    sub = np.ix_(mask, mask)
    np.testing.assert_allclose(H[sub], Ho[sub], rtol=0, atol=1e-10 * np.abs(Ho[sub]).max())
    np.testing.assert_allclose(b[mask], bo[mask], rtol=0, atol=1e-10 * np.abs(bo[mask]).max())
    chi2, H2, b2 = gr.linearize()                                  # the same launches again: bitwise
    assert chi2 == chi and np.array_equal(H, H2) and np.array_equal(b, b2)


def test_imu_factors_appended_in_place_match_a_rebuild():
    """The structure's in-place growth (refresh_factors) colours the IMU factors of the new keyframes and uploads the
    colour-sorted list again.  A VIO graph grown keyframe by keyframe through fgo_isam2_update (threshold never reached: the
    linearisation point stays) must give the same H / b / chi2 as the same context after a structure rebuild."""
    rng = np.random.default_rng(11)
    g = same_information(vio_graph(rng, n_kf=16, with_planes=False))
    K, n0 = g["n_kf"], 6
    newest = np.maximum(g["ei"], g["ej"])
    gr = G.Graph()

    def add_keyframe(k):
        gr.add_poses(g["values"][k:k + 1], ids=[k])
        gr.add_vec3(K + k, g["values"][K + k, :3])
        gr.add_bias(2 * K + k, g["values"][2 * K + k, :6])
        for e in np.nonzero(newest == k)[0]:
            assert g["kind"][e] == orc.FK_BETWEEN
            gr.add_edges([int(g["ei"][e])], [int(g["ej"][e])], g["meas"][e:e + 1], g["info"][e:e + 1], tangent_order=G.FGO_TANGENT_GTSAM)
        if k > 0:
            assert max(g["imu_ids"][k - 1]) == 2 * K + k
            gr.add_imu(g["imu_ids"][k - 1], g["imu_pre"][k - 1].buf)

    for k in range(n0):
        add_keyframe(k)
    gr.add_prior(0, g["prior_mean"][0], g["prior_info"][0])
    gr.add_prior_vec3(K, g["prior_mean"][1][:3], 1e-3)
    gr.add_prior_bias(2 * K, g["prior_mean"][2][:6], 1e-3)
    stats = [gr.isam2_update(1e9)]
    for k in range(n0, K):
        add_keyframe(k)
        stats.append(gr.isam2_update(1e9))
    assert sum(st.structure_rebuilt == 0 for st in stats[1:]) >= (K - n0) // 2, [st.structure_rebuilt for st in stats]
    chi, H, b = gr.linearize()
    gr.isam2_reset()                                               # the next use rebuilds the structure (and the colours) from scratch
    chi2, H2, b2 = gr.linearize()
    assert H.shape == H2.shape == (18 * K, 18 * K)
    assert abs(chi - chi2) <= 1e-12 * chi2
    np.testing.assert_allclose(H, H2, rtol=0, atol=1e-12 * np.abs(H2).max())
    np.testing.assert_allclose(b, b2, rtol=0, atol=1e-12 * np.abs(b2).max())


def test_preint_information_is_the_inverse_covariance():
    rng = np.random.default_rng(5)
    g = vio_graph(rng, n_kf=4, with_planes=False)
    for p in g["imu_pre"]:
        W = G.preint_information(p.buf)
        np.testing.assert_array_equal(W, W.T)
        np.testing.assert_allclose(W @ p.cov, np.eye(15), atol=1e-6)
        np.testing.assert_allclose(W, np.linalg.inv(p.cov), rtol=0, atol=1e-6 * np.abs(W).max())


@pytest.mark.parametrize("seed,planes", [(0, False), (1, True)])
def test_vio_linearization_independent_inverse(seed, planes):
    """the oracle weighted with numpy.linalg.inv(preintMeasCov) instead: agreement bounded by cond(cov) ~ 1e9"""
    rng = np.random.default_rng(seed)
    g = vio_graph(rng, n_kf=7, with_planes=planes)
    gr, po = vio_gpu(g), mixed_oracle(g)
    chi, H, b = gr.linearize()
    Ho, bo = po.dense_system()
    assert abs(chi - po.chi2()) <= 1e-6 * po.chi2()
    mask = np.ones(len(bo), bool); mask[:6] = False
    sub = np.ix_(mask, mask)
    np.testing.assert_allclose(H[sub], Ho[sub], rtol=0, atol=1e-6 * np.abs(Ho[sub]).max())
    np.testing.assert_allclose(b[mask], bo[mask], rtol=0, atol=1e-6 * np.abs(bo[mask]).max())
    assert abs(gr.chi2() - po.chi2()) <= 1e-6 * po.chi2()        # stand-alone chi2 kernel (IMU part included)


def test_vio_lm_matches_oracle():
    rng = np.random.default_rng(2)
    g = same_information(vio_graph(rng, n_kf=10, with_planes=True))
    gr, po = vio_gpu(g), mixed_oracle(g)
    rg, sg = gr.optimize_gtsam()
    ro, so = po.optimize_gtsam()
    assert rg == ro and sg.trials == so.trials
    np.testing.assert_allclose(gr.trace()[1], po.trace()[1], rtol=1e-12)
    # the 1e14 prior on X0 (sigma 1e-7, gtsam_graph.cpp:338-347) puts cond(H) near 1e16 / smallest pivot: the two Cholesky
    # orderings agree to ~1e-8 on the error and ~1e-7 on the values after ~10 iterations
    assert abs(gr.error() - po.error_gtsam()) <= 1e-7 * po.error_gtsam()
    V, Vo = gr.get_poses(), po.get_poses()
    K = g["n_kf"]
    assert np.abs(V[:K, :3] - Vo[:K, :3]).max() < 1e-6
    assert np.abs(V[K:2 * K, :3] - Vo[K:2 * K, :3]).max() < 1e-6          # velocities
    assert np.abs(V[2 * K:3 * K, :6] - Vo[2 * K:3 * K, :6]).max() < 1e-7  # biases
    assert sg.chi2_final < 1e-2 * sg.chi2_initial


def test_vio_determinism_and_repeat_linearization():
    """the IMU kernel accumulates (+=) into H: two consecutive linearisations at the same point must agree bitwise"""
    rng = np.random.default_rng(3)
    g = vio_graph(rng, n_kf=6, with_planes=False)
    gr = vio_gpu(g)
    c1, H1, b1 = gr.linearize()
    c2, H2, b2 = gr.linearize()
    assert c1 == c2
    np.testing.assert_array_equal(H1, H2); np.testing.assert_array_equal(b1, b2)


def test_imu_key_type_checks():
    gr = G.Graph()
    gr.add_poses(np.array([[0, 0, 0, 0, 0, 0, 1.0]] * 2))
    gr.add_vec3(10, [0, 0, 0]); gr.add_vec3(11, [0, 0, 0]); gr.add_bias(20, np.zeros(6)); gr.add_bias(21, np.zeros(6))
    pim = G.Preintegrator()
    with pytest.raises(G.FgoError):
        gr.add_imu([0, 10, 1, 11, 20, 21], pim.buf)              # empty preintegration
    pim.integrate([0, 0, -9.71], [0, 0, 0], 0.005)
    with pytest.raises(G.FgoError):
        gr.add_imu([0, 1, 10, 11, 20, 21], pim.buf)              # wrong key order
    gr.add_imu([0, 10, 1, 11, 20, 21], pim.buf)


def test_preint_batch_matches_host_preintegrator():
    """fgo_preint_batch (one wave per factor on the GPU) vs the host loop of fgo_preint_integrate, ragged sample counts"""
    rng = np.random.default_rng(3)
    counts = np.array([40, 1, 0, 17, 200, 40, 40, 3])
    sp = np.concatenate([[0], np.cumsum(counts)])
    acc = rng.normal(size=(sp[-1], 3)) * 0.5 + np.array([0, 0, -9.7])
    gyro = rng.normal(size=(sp[-1], 3)) * 0.3
    bias = rng.normal(size=(len(counts), 6)) * 0.02
    out = G.preint_batch(sp, acc, gyro, 0.005, bias_hat=bias)
    for f in range(len(counts)):
        pim = G.Preintegrator(bias_hat=bias[f])
        for s in range(sp[f], sp[f + 1]):
            pim.integrate(acc[s], gyro[s], 0.005)
        ref = pim.buf
        scale = np.maximum(np.abs(ref), 1e-12)
        assert np.all(np.abs(out[f] - ref) <= 1e-11 * np.maximum(scale, np.abs(ref).max() * 1e-6)), (f, np.abs(out[f] - ref).max())
    # the factor built from a batched payload behaves like the one built from the host payload
    out0 = G.preint_batch(sp, acc, gyro, 0.005)
    pim = G.Preintegrator()
    for s in range(sp[0], sp[1]):
        pim.integrate(acc[s], gyro[s], 0.005)
    np.testing.assert_allclose(out0[0], pim.buf, rtol=1e-10, atol=1e-14)


def test_preint_batch_matches_oracle():
    """fgo_preint_batch (csrc/preint_kernel.hip, one wave per factor) against the ORACLE's restatement of
    PreintegratedCombinedMeasurements (oracle/orc_imu.h, orc.Preint) -- not against the product's own host loop.
    Ragged sample counts incl. an empty factor; tolerance: deltas / bias Jacobians 1e-12 absolute, covariance 1e-10
    relative to its largest entry."""
    rng = np.random.default_rng(11)
    counts = np.array([40, 1, 0, 17, 200, 40, 3, 64, 65, 400])
    sp = np.concatenate([[0], np.cumsum(counts)])
    acc = rng.normal(size=(sp[-1], 3)) * 0.5 + np.array([0, 0, -9.7])
    gyro = rng.normal(size=(sp[-1], 3)) * 0.3
    bias = rng.normal(size=(len(counts), 6)) * 0.02
    out = G.preint_batch(sp, acc, gyro, 0.005, bias_hat=bias)
    for f in range(len(counts)):
        ref = orc.Preint(bias[f], acc[sp[f]:sp[f + 1]].reshape(-1, 3), gyro[sp[f]:sp[f + 1]].reshape(-1, 3), 0.005)
        assert len(ref.buf) == out.shape[1]
        np.testing.assert_allclose(out[f, :62], ref.buf[:62], rtol=0, atol=1e-12)
        cmax = max(np.abs(ref.buf[62:]).max(), 1e-300)
        np.testing.assert_allclose(out[f, 62:], ref.buf[62:], rtol=1e-10, atol=1e-10 * cmax)
        if counts[f] == 0:
            continue
        # and the factor evaluated by the oracle from the GPU payload == from the oracle payload
        xi = np.concatenate([rng.normal(size=3), [0, 0, 0, 1.0]]); vi = rng.normal(size=3)
        gpu = orc.Preint(bias[f], np.zeros((0, 3)), np.zeros((0, 3)), 0.005); gpu.buf[:] = out[f]
        xj, vj = ref.predict(xi, vi, bias[f] + 1e-3)
        xg, vg = gpu.predict(xi, vi, bias[f] + 1e-3)
        np.testing.assert_allclose(xg, xj, atol=1e-11); np.testing.assert_allclose(vg, vj, atol=1e-11)


def test_imu_factor_with_a_repeated_variable_is_rejected():
    """k_imu_blocks adds a factor's 21 blocks with one lane each, so a CombinedImuFactor that lists a variable
    twice (B_i == B_j, X_i == X_j) would lose contributions; the entry point refuses it (GTSAM's factor has six distinct keys)"""
    rng = np.random.default_rng(2)
    g = vio_graph(rng, n_kf=4, with_planes=False)
    gr = vio_gpu(g)
    K = g["n_kf"]
    ids = np.array([0, K, 1, K + 1, 2 * K, 2 * K], np.int64)
    with pytest.raises(G.FgoError):
        gr.add_imu(ids, g["imu_pre"][0].buf)
    ids = np.array([0, K, 0, K + 1, 2 * K, 2 * K + 1], np.int64)
    with pytest.raises(G.FgoError):
        gr.add_imu(ids, g["imu_pre"][0].buf)
    rc, st = gr.optimize_gtsam(3)                                   # the graph is untouched by the refused calls
    assert rc >= 1
