"""A second, independently written evaluation of the GTSAM Pose3 definitions the reference's GTSAM path rests on -- CHECKER SIDE ONLY.

SURVEY.md Appendix A.2 from the prose again, sharing no formula with oracle/orc_pose3.h or csrc/pose3_device.hpp: a pose is a 4x4 matrix, the
tangent [omega; v] maps to the twist matrix [[ [omega]x, v ], [0, 0]], Expmap / Logmap are the MATRIX exponential / logarithm computed by mpmath
at 40 digits (Pade / Schur inside the library -- no Rodrigues formula, no closed-form SE(3) Jacobians here), retract(X, d) = X Expmap(d),
  PriorFactor<Pose3>    r = Logmap(prior^-1 X)                      gtsam/gtsam_graph.cpp:338-341
  BetweenFactor<Pose3>  r = Logmap(Z^-1 Xi^-1 Xj)                   gtsam/gtsam_graph.cpp:689-692
and every Jacobian is a central difference of exactly those definitions in 40-digit arithmetic (step 1e-12: truncation 1e-24, rounding 1e-28),
i.e. exact to double precision without a single derived formula."""
import mpmath as mp
import numpy as np

mp.mp.dps = 40


def hat(d):
    w, v = d[:3], d[3:]
    return mp.matrix([[0, -w[2], w[1], v[0]], [w[2], 0, -w[0], v[1]], [-w[1], w[0], 0, v[2]], [0, 0, 0, 0]])


def vee(M):
    return [M[2, 1], M[0, 2], M[1, 0], M[0, 3], M[1, 3], M[2, 3]]


def pose_mat(p):
    """tx ty tz qx qy qz qw -> 4x4 (mpmath), the rotation as (w^2 - v.v) I + 2 v v^T + 2 w [v]x"""
    q = [mp.mpf(float(x)) for x in p[3:]]
    nq = mp.sqrt(sum(x * x for x in q))
    x, y, z, w = [c / nq for c in q]
    v = [x, y, z]
    vv = x * x + y * y + z * z
    S = [[0, -z, y], [z, 0, -x], [-y, x, 0]]
    T = mp.eye(4)
    for r in range(3):
        for c in range(3):
            T[r, c] = 2 * v[r] * v[c] + 2 * w * S[r][c] + ((w * w - vv) if r == c else 0)
        T[r, 3] = mp.mpf(float(p[r]))
    return T


def inv(T):
    return mp.inverse(T)


def expmap(d):
    return mp.expm(hat([mp.mpf(x) for x in d]))


def logmap(T):
    L = mp.logm(T)
    return [mp.re(x) for x in vee(L)]


def between_residual(Xi, Xj, Z, di=None, dj=None):
    A = Xi * expmap(di) if di is not None else Xi
    B = Xj * expmap(dj) if dj is not None else Xj
    return logmap(inv(Z) * inv(A) * B)


def prior_residual(X, M, d=None):
    A = X * expmap(d) if d is not None else X
    return logmap(inv(M) * A)


def _jac(fn, h=mp.mpf("1e-12")):
    cols = []
    for k in range(6):
        dp = [mp.mpf(0)] * 6; dm = [mp.mpf(0)] * 6
        dp[k] = h; dm[k] = -h
        fp, fm = fn(dp), fn(dm)
        cols.append([(a - b) / (2 * h) for a, b in zip(fp, fm)])
    return np.array([[float(cols[c][r]) for c in range(6)] for r in range(6)])


def between(xi, xj, z):
    """r (6), Ji, Jj (6x6) as floats"""
    Xi, Xj, Z = pose_mat(xi), pose_mat(xj), pose_mat(z)
    r = np.array([float(x) for x in between_residual(Xi, Xj, Z)])
    Ji = _jac(lambda d: between_residual(Xi, Xj, Z, di=d))
    Jj = _jac(lambda d: between_residual(Xi, Xj, Z, dj=d))
    return r, Ji, Jj


def prior(x, mean):
    X, M = pose_mat(x), pose_mat(mean)
    r = np.array([float(v) for v in prior_residual(X, M)])
    return r, _jac(lambda d: prior_residual(X, M, d))


def retract(x, d):
    """X Expmap(d) as tx ty tz qx qy qz qw (w >= 0), the quaternion from the rotation matrix by its largest-diagonal branch"""
    T = pose_mat(x) * expmap(d)
    R = np.array([[float(T[r, c]) for c in range(3)] for r in range(3)])
    t = np.array([float(T[r, 3]) for r in range(3)])
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = 2.0 * np.sqrt(tr + 1.0)
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, s / 4.0])
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]])); j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i] = s / 4.0; q[j] = (R[j, i] + R[i, j]) / s; q[k] = (R[k, i] + R[i, k]) / s; q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    return np.concatenate([t, q / np.linalg.norm(q)])
