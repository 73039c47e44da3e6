#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (the reference holds no golden vectors for the
pose-graph, BA or IMU paths, so the fixtures are produced by this build's CPU oracle on tiny graphs and committed;
they pin the HIP kernels to the ORACLE, not to g2o / GTSAM -- "parity unpinned" stays true for those paths).

    python tests/golden/make_golden.py            # rewrite the fixtures
    python tests/golden/make_golden.py --check    # recompute and compare with the committed files (what the CPU test does)

Fixtures (numpy .npz, inputs + expected outputs):
  se3_triangle3 / se3_ring10 / se3_manhattan100   g2o semantics (g2o_graph.cpp:65-134, 241-258): per-edge error and
        Jacobians (EdgeSE3), chi2, dense H and b (small ones), the undamped step, chi2 / lambda after every LM
        iteration of the reference's schedule 10 x optimize(2), final poses
  gtsam_chain12      GTSAM semantics (gtsam_graph.cpp:338-341, 640-692, 1784-1788): prior + between factors; error,
        dense H / b, LevenbergMarquardtOptimizer trajectory, final poses, three ISAM2 steps
  gtsam_mixed        poses + plane landmarks + points (gtsam_graph.cpp:373-409, 1118-1298): error, H / b, LM trajectory
Nothing here needs a GPU; the synthetic Manhattan generator is the host-only fgo_synth_manhattan3d."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from tests import orc_binding as orc
from tests.util import small_graph, info_ut, mixed_graph, mixed_oracle, pose_mul, pose_inv, noisy, random_info


def se3_case(g):
    """g2o-semantics pose graph -> dict of inputs and oracle outputs"""
    out = {k: np.asarray(g[k]) for k in ("poses", "fixed", "ei", "ej", "meas", "info")}
    E = len(g["ei"])
    e = np.zeros((E, 6)); Ji = np.zeros((E, 6, 6)); Jj = np.zeros((E, 6, 6))
    for k in range(E):
        e[k], Ji[k], Jj[k] = orc.edge_se3(g["poses"][g["ei"][k]], g["poses"][g["ej"][k]], g["meas"][k])
    out["edge_error"], out["edge_Ji"], out["edge_Jj"] = e, Ji, Jj
    po = orc.Problem(g["poses"], g["fixed"], g["ei"], g["ej"], g["meas"], g["info"])
    out["chi2_initial"] = np.array(po.chi2())
    H, b = po.dense_system()
    if H.shape[0] <= 120:
        out["H"] = H
    out["b"] = b
    rc, d = po.solve_step(0.0)
    assert rc == 0
    out["step_undamped"] = d
    chi_calls, tr_chi, tr_lam, iters = [], [], [], []
    for call in range(10):                      # CGraphG2O::optimizeGraph, g2o_graph.cpp:241-252
        rc, st = po.optimize(2)
        iters.append(rc); chi_calls.append(po.chi2())
        c, l = po.trace()
        tr_chi.extend(c); tr_lam.extend(l)
    out["iterations_per_call"] = np.array(iters); out["chi2_after_call"] = np.array(chi_calls)
    out["trace_chi2"] = np.array(tr_chi); out["trace_lambda"] = np.array(tr_lam)
    out["poses_final"] = po.get_poses()
    return out


def triangle3():
    rng = np.random.default_rng(101)
    g = small_graph(rng, n=3, extra=0, noise=0.03)
    # close the triangle
    from tests.util import pose_mul, pose_inv, noisy, random_info
    z = noisy(rng, pose_mul(pose_inv(g["poses"][0]), g["poses"][2]), 0.02, 0.01)
    g["ei"] = np.append(g["ei"], 0).astype(np.int32); g["ej"] = np.append(g["ej"], 2).astype(np.int32)
    g["meas"] = np.vstack([g["meas"], z]); g["info"] = np.vstack([g["info"], info_ut(random_info(rng))])
    return g


def ring10():
    rng = np.random.default_rng(102)
    n = 10
    truth = []
    for k in range(n):
        a = 2 * np.pi * k / n
        truth.append(np.array([3 * np.cos(a), 3 * np.sin(a), 0.2 * k, 0, 0, np.sin((a + np.pi / 2) / 2), np.cos((a + np.pi / 2) / 2)]))
    truth = np.array(truth)
    pairs = [(k, (k + 1) % n) for k in range(n)] + [(0, 5), (2, 7), (3, 8)]
    pairs = [(min(a, b), max(a, b)) for a, b in pairs]
    meas = np.array([noisy(rng, pose_mul(pose_inv(truth[a]), truth[b]), 0.02, 0.01) for a, b in pairs])
    info = np.array([info_ut(random_info(rng)) for _ in pairs])
    poses = np.array([noisy(rng, t, 0.1, 0.03) for t in truth]); poses[0] = truth[0]
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    return dict(poses=poses, fixed=fixed, ei=np.array([p[0] for p in pairs], np.int32), ej=np.array([p[1] for p in pairs], np.int32),
                meas=meas, info=info)


def manhattan100():
    import graph_slam_amd as G      # host-only generator; the library is loaded but no device call is made
    g = G.synth_manhattan3d(100, 4, 2, seed=42)
    fixed = np.zeros(100, np.uint8); fixed[0] = 1
    return dict(poses=g["poses"], fixed=fixed, ei=g["ei"].astype(np.int32), ej=g["ej"].astype(np.int32), meas=g["meas"], info=g["info"])


PRIOR = info_ut(np.diag([1e6] * 6))


def gtsam_chain12():
    rng = np.random.default_rng(103)
    g = small_graph(rng, n=12, extra=8, noise=0.03, fixed_first=False)
    out = {k: np.asarray(g[k]) for k in ("poses", "ei", "ej", "meas", "info")}
    out["prior_info"] = PRIOR

    def problem():
        po = orc.Problem(g["poses"], np.zeros(12, np.uint8), g["ei"], g["ej"], g["meas"], g["info"])
        po.set_gtsam()
        po.add_priors(np.array([0], np.int32), g["poses"][:1], PRIOR[None, :])
        return po
    po = problem()
    out["error_initial"] = np.array(po.error_gtsam())
    H, b = po.dense_system()
    out["H"], out["b"] = H, b
    rc, st = po.optimize_gtsam()
    out["lm_iterations"] = np.array(rc); out["lm_trials"] = np.array(st.trials)
    c, l = po.trace()
    out["trace_chi2"], out["trace_lambda"] = c, l
    out["poses_final"] = po.get_poses(); out["error_final"] = np.array(po.error_gtsam())
    po = problem()                                                # ISAM2 semantics, threshold 0.05, three updates
    theta = np.ascontiguousarray(g["poses"].copy()); delta = np.zeros((12, 6)); est = []
    moved = []
    for _ in range(3):
        e, m = po.isam2_step(0.05, theta, delta)
        est.append(e); moved.append(m)
    out["isam2_estimates"] = np.array(est); out["isam2_relinearised"] = np.array(moved)
    out["isam2_theta"], out["isam2_delta"] = theta, delta
    return out


def gtsam_mixed():
    rng = np.random.default_rng(104)
    g = mixed_graph(rng, n_poses=6, n_planes=2, n_points=8)
    out = {k: np.asarray(v) for k, v in g.items()}
    po = mixed_oracle(g)
    out["error_initial"] = np.array(po.error_gtsam())
    H, b = po.dense_system()
    out["H"], out["b"] = H, b
    rc, st = po.optimize_gtsam()
    out["lm_iterations"] = np.array(rc); out["lm_trials"] = np.array(st.trials)
    c, l = po.trace()
    out["trace_chi2"], out["trace_lambda"] = c, l
    out["values_final"] = po.get_poses(); out["error_final"] = np.array(po.error_gtsam())
    return out


CASES = {"se3_triangle3": lambda: se3_case(triangle3()), "se3_ring10": lambda: se3_case(ring10()),
         "se3_manhattan100": lambda: se3_case(manhattan100()), "gtsam_chain12": gtsam_chain12, "gtsam_mixed": gtsam_mixed}


# Pins no tighter than the arithmetic reproduces across compilers / ISA levels: whether gcc contracts
# a*b+c into an FMA, and how wide it vectorises a reduction, moves g2o's lambda (a continuous function of rho) in the 11th
# digit on the tiny gauge-sensitive graphs, and everything downstream of it accordingly.  Single-pass quantities (chi2, H, b,
# Jacobians) keep 1e-10; the LM trajectories carry the tolerances the GPU-side tests use for the same keys.
KEY_RTOL = {"trace_lambda": 1e-6, "trace_chi2": 1e-8, "chi2_after_call": 1e-8, "poses_final": 1e-7, "values_final": 1e-6,
            "lambda_final": 1e-6, "step_undamped": 1e-8, "isam2_estimates": 1e-8, "isam2_theta": 1e-8, "isam2_delta": 1e-8,
            "error_final": 1e-6}


def compare(name, new, old, rtol=1e-10):
    bad = []
    for k in new:
        if k not in old.files:
            bad.append("%s: missing key %s" % (name, k)); continue
        a, b = np.asarray(new[k], dtype=np.float64), np.asarray(old[k], dtype=np.float64)
        rt = max(rtol, KEY_RTOL.get(k, 0.0))
        if a.shape != b.shape or not np.allclose(a, b, rtol=rt, atol=rt * max(1.0, float(np.abs(b).max()) if b.size else 1.0)):
            bad.append("%s: %s differs" % (name, k))
    return bad


if __name__ == "__main__":
    check = "--check" in sys.argv
    problems = []
    for name, fn in CASES.items():
        data = fn()
        path = os.path.join(HERE, name + ".npz")
        if check:
            problems += compare(name, data, np.load(path))
        else:
            np.savez_compressed(path, **data)
            print("wrote", path, os.path.getsize(path), "bytes")
    if check:
        print("\n".join(problems) if problems else "golden fixtures reproduce")
        sys.exit(1 if problems else 0)
