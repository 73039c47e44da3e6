"""The panel (supernode) kernels at the top of the elimination tree -- MFMA triangle / row kernels, fused forward solve,
tile-based backward solve -- against dense numpy solves and against the generic kernels, over schedules that force
panels of every width (tiny light-subtree limits make almost every column part of a panel; FGO_NO_PANELS disables them).
Edge cases covered by the sizes: panels of 1..16 columns (6 m not a multiple of the 16-wide tiles), root panels without
off-triangle rows, panels whose rows spill over several 16-row MFMA chunks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import graph_slam_amd as G
from tests.test_gpu_parity import synth, make_gpu, make_orc


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("n,task_work", [(40, 1), (150, 1), (150, 30), (400, 1), (400, 200), (650, 50), (650, 5000)])
def test_damped_solve_matches_dense(n, task_work):
    """stand-alone factor + forward (k_fwd_ext / k_fwd_tri) + backward (k_bwd_ext / k_bwd_tri) vs numpy"""
    g = synth(n, 5, 4, seed=100 + n)

    def run():
        gr = make_gpu(g)
        chi, H, b = gr.linearize(dense=True)
        lam = 1e-5 * np.abs(np.diag(H)).max()
        d = gr.solve_step(lam)
        return H, b, lam, d, gr.stats()
    H, b, lam, d, st = with_env({"FGO_TASK_WORK": task_work, "FGO_NO_PANELS": None}, run)
    ref = np.linalg.solve(H + lam * np.eye(len(b)), b)
    np.testing.assert_allclose(d, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    if task_work <= 50 and n >= 150:
        assert st.n_levels > 3     # the schedule really is panel-dominated


@pytest.mark.parametrize("n,task_work", [(300, 1), (3000, 40), (3000, 2000), (20000, 5000)])
def test_lm_with_panels_matches_generic_kernels_and_oracle(n, task_work):
    """full LM (fused forward solve in the factor sweep) with panels == without panels == oracle"""
    g = synth(n, 5, 4, seed=7 + n)

    def run():
        gr = make_gpu(g)
        rc, st = gr.optimize(4)
        return rc, np.array(gr.trace()[0]), gr.get_poses().copy()
    rc_p, tr_p, x_p = with_env({"FGO_TASK_WORK": task_work, "FGO_NO_PANELS": None}, run)
    rc_g, tr_g, x_g = with_env({"FGO_TASK_WORK": task_work, "FGO_NO_PANELS": 1}, run)
    assert rc_p == rc_g
    np.testing.assert_allclose(tr_p, tr_g, rtol=1e-10)
    np.testing.assert_allclose(x_p, x_g, atol=1e-8)
    if n <= 3000:
        po = make_orc(g)
        ro, so = po.optimize(4)
        assert ro == rc_p
        np.testing.assert_allclose(tr_p[-1], so.chi2_final, rtol=1e-9)
        np.testing.assert_allclose(x_p[:, :3], po.get_poses()[:, :3], atol=1e-7)


def test_not_positive_definite_is_reported():
    """a negative-definite information matrix must surface as FGO_ENUM from the panel path too"""
    g = synth(200, 3, 1, seed=5)
    g["info"] = -np.asarray(g["info"])

    def run():
        gr = make_gpu(g)
        with pytest.raises(Exception):
            gr.solve_step(0.0)
    with_env({"FGO_TASK_WORK": 1}, run)


def test_throughput_triangle_kernel_and_unchained_backward_sweep_on_every_level():
    """k_panel_tri1 (one wave per panel, the form the wide levels run) normally only sees levels with >= 768 panels, i.e. graphs
    of ~30 000 poses and more.  FGO_TUNE="tri_wide=0,tri1_min=0" sends EVERY panel level of the small graphs above through it --
    panels of 1 .. 16 columns, 6 m not a multiple of the tile width, against numpy and the oracle -- and FGO_TUNE="bwd_chain=0" runs
    the same tests on per-level backward launches instead of the chained one (the launch-selection variables are read once per
    process, hence the child process)."""
    import subprocess
    import sys
    if os.environ.get("FGO_PANEL_VARIANTS_CHILD"):
        pytest.skip("child run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"FGO_TUNE": "tri_wide=0,tri1_min=0"}, {"FGO_TUNE": "bwd_chain=0"}):
        env = dict(os.environ, FGO_PANEL_VARIANTS_CHILD="1", **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_panels.py"), "-q", "-x", "-m", "gpu",
                            "-k", "damped or lm_with", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (extra, r.stdout[-2000:], r.stderr[-2000:])
