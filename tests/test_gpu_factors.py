"""GPU parity tests of the plane (OrientedPlane3Factor) and reprojection (GenericProjectionFactor + Cal3DS2) kernels
and of mixed-variable graphs through the C-ABI, against the oracle; plus the reference's own plane-fusion known
answers (gtsam/test/testOrientedPlane3Factor.cpp:37-126) solved end to end on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from tests import orc_binding as orc
from tests.util import mixed_graph, mixed_oracle, info_ut


def mixed_gpu(g):
    gr = G.Graph()
    n_poses, n_planes = g["n_poses"], g["n_planes"]
    gr.add_poses(g["values"][:n_poses])
    for v in range(n_poses, n_poses + n_planes):
        gr.add_plane(v, g["values"][v, :4])
    for v in range(n_poses + n_planes, len(g["values"])):
        gr.add_point(v, g["values"][v, :3])
    gr.set_calibration(g["calib"], g["bps"])
    for k in range(len(g["ei"])):
        i, j, kind = int(g["ei"][k]), int(g["ej"][k]), g["kind"][k]
        if kind == orc.FK_BETWEEN:
            gr.add_edges([i], [j], g["meas"][k:k + 1], g["info"][k:k + 1], tangent_order=G.FGO_TANGENT_GTSAM)
        elif kind == orc.FK_PLANE:
            # the C-ABI takes the covariance (Gaussian::Covariance(S)); the scenario stores information diag(1e4)
            gr.add_plane_factor(i, j, g["meas"][k, :4], [1e-4, 0, 0, 1e-4, 0, 1e-4])
        else:
            gr.add_reproj(i, j, g["meas"][k, :2], 1.0)
    for q, v in enumerate(g["prior_ids"]):
        if g["vkind"][v] == orc.VK_POSE:
            gr.add_prior(int(v), g["prior_mean"][q], g["prior_info"][q])
        else:
            gr.add_prior_point(int(v), g["prior_mean"][q, :3], 0.014)
    return gr


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mixed_graph_linearization_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    g = mixed_graph(rng, n_poses=10, n_planes=4, n_points=20)
    gr, po = mixed_gpu(g), mixed_oracle(g)
    chi, H, b = gr.linearize()
    Ho, bo = po.dense_system()
    assert abs(chi - po.chi2()) <= 1e-10 * po.chi2()
    mask = np.ones(len(bo), bool); mask[:6] = False                 # the 1e14 prior block (see test_gpu_gtsam)
    sub = np.ix_(mask, mask)
    np.testing.assert_allclose(H[sub], Ho[sub], rtol=0, atol=1e-10 * np.abs(Ho[sub]).max())
    np.testing.assert_allclose(H, Ho, rtol=0, atol=1e-12 * np.abs(Ho).max())
    np.testing.assert_allclose(b[mask], bo[mask], rtol=0, atol=1e-10 * np.abs(bo[mask]).max())


def test_mixed_graph_lm_matches_oracle():
    rng = np.random.default_rng(5)
    g = mixed_graph(rng, n_poses=12, n_planes=4, n_points=30)
    gr, po = mixed_gpu(g), mixed_oracle(g)
    rg, sg = gr.optimize_gtsam()
    ro, so = po.optimize_gtsam()
    assert rg == ro and sg.trials == so.trials
    np.testing.assert_allclose(gr.trace()[1], po.trace()[1], rtol=1e-12)
    np.testing.assert_allclose(gr.trace()[0], po.trace()[0], rtol=1e-6)
    assert abs(gr.error() - po.error_gtsam()) <= 1e-6 * po.error_gtsam()
    V, Vo = gr.get_poses(), po.get_poses()
    np_, npl = g["n_poses"], g["n_planes"]
    sgn = np.sign(np.sum(V[:np_, 3:] * Vo[:np_, 3:], axis=1))[:, None]
    assert np.abs(V[:np_, :3] - Vo[:np_, :3]).max() < 1e-6
    assert np.abs(V[:np_, 3:] * sgn - Vo[:np_, 3:]).max() < 1e-6
    np.testing.assert_allclose(V[np_:np_ + npl, :4], Vo[np_:np_ + npl, :4], atol=1e-6)        # planes
    np.testing.assert_allclose(V[np_ + npl:, :3], Vo[np_ + npl:, :3], atol=1e-6)              # points
    assert np.abs(np.linalg.norm(V[np_:np_ + npl, :3], axis=1) - 1).max() < 1e-12             # unit normals


def _fusion(meas):
    """one pose (prior sigma 1e-3) + one plane landmark + two plane measurements (sigma 0.1), as in the reference's
    copied GTSAM tests; the optimum of that tiny problem is the known answer"""
    gr = G.Graph()
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    gr.add_poses(ident[None, :])
    gr.add_prior(0, ident, info_ut(np.eye(6) / 1e-3 ** 2))
    gr.add_plane(100, [-1.0, 0.0, 0.0, 3.0])
    for z in meas:
        gr.add_plane_factor(0, 100, z, [0.01, 0, 0, 0.01, 0, 0.01])
    gr.optimize_gtsam()
    return gr.get_poses(ids=[100])[0, :4]


def test_reference_golden_plane_fusion_range():
    np.testing.assert_allclose(_fusion([[-1.0, 0, 0, 3.0], [-1.0, 0, 0, 1.0]]), [-1, 0, 0, 2.0], atol=1e-7)


def test_reference_golden_plane_fusion_angle():
    s = np.sqrt(2) / 2
    np.testing.assert_allclose(_fusion([[-1.0, 0, 0, 3.0], [0, -1.0, 0, 3.0]]), [-s, -s, 0, 3.0], atol=1e-7)


def test_marginal_covariance_blocks():
    """Marginals::marginalCovariance (gtsam_graph.cpp:598-601): diagonal blocks of (J' Omega J)^-1 from the resident
    factor vs numpy.linalg.inv of the oracle's dense information matrix.  Relative tolerance 1e-8 per block."""
    from tests.util import small_graph
    rng = np.random.default_rng(8)
    g = small_graph(rng, n=30, extra=40)                      # g2o semantics, vertex 0 fixed
    gr = G.Graph(); gr.add_poses(g["poses"], g["fixed"]); gr.add_edges(g["ei"], g["ej"], g["meas"], g["info"])
    po = orc.Problem(g["poses"], g["fixed"], g["ei"], g["ej"], g["meas"], g["info"])
    Hinv = np.linalg.inv(po.dense_system()[0])
    for v in (1, 7, 29):
        blk = Hinv[6 * (v - 1):6 * v, 6 * (v - 1):6 * v]      # free-variable order = ascending id, pose 0 fixed
        np.testing.assert_allclose(gr.marginal_cov(v), blk, rtol=0, atol=1e-8 * np.abs(blk).max())
    with pytest.raises(G.FgoError):
        gr.marginal_cov(0)                                    # fixed vertex
    # several blocks from one resident factorisation; the factor survives between calls and is refreshed after a change
    many = gr.marginal_cov_many([29, 1, 7, 7])
    for q, v in enumerate((29, 1, 7, 7)):
        blk = Hinv[6 * (v - 1):6 * v, 6 * (v - 1):6 * v]
        np.testing.assert_allclose(many[q], blk, rtol=0, atol=1e-8 * np.abs(blk).max())
    gr.optimize(1)
    po.optimize(1)
    Hinv_b = np.linalg.inv(po.dense_system()[0])
    blk = Hinv_b[6 * 6:6 * 7, 6 * 6:6 * 7]
    np.testing.assert_allclose(gr.marginal_cov_many([7])[0], blk, rtol=0, atol=1e-7 * np.abs(blk).max())
    g2 = mixed_graph(np.random.default_rng(9), n_poses=8, n_planes=3, n_points=10)
    gr2, po2 = mixed_gpu(g2), mixed_oracle(g2)
    Hinv2 = np.linalg.inv(po2.dense_system()[0])
    for v in (3, g2["n_poses"] + 1, g2["n_poses"] + g2["n_planes"] + 2):     # a pose, a plane, a point
        blk = Hinv2[6 * v:6 * v + 6, 6 * v:6 * v + 6]
        np.testing.assert_allclose(gr2.marginal_cov(v), blk, rtol=0, atol=1e-7 * np.abs(blk).max())


def test_type_checks():
    gr = G.Graph()
    gr.add_poses(np.array([[0, 0, 0, 0, 0, 0, 1.0]]))
    gr.add_point(7, [1, 2, 3])
    with pytest.raises(G.FgoError):
        gr.add_plane_factor(0, 7, [1, 0, 0, 1], [1, 0, 0, 1, 0, 1])        # 7 is a point, not a plane
    with pytest.raises(G.FgoError):
        gr.add_reproj(0, 7, [1, 2]); gr.chi2()                             # no calibration set
