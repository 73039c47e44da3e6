"""CombinedImuFactor a second time -- CHECKER SIDE ONLY (round 6).  From the definitions alone (GTSAM 4.0's
PreintegrationBase::computeError: the error is state_j.localCoordinates(predicted state_j) of NavStates, CombinedImuFactor appends Between(bias_j,
bias_i)), with 3x3 / 4x4 mpmath matrices at 40 digits, the rotation logarithm as a MATRIX logarithm, and every Jacobian a central difference of
exactly that code: no closed-form SO(3) Jacobian, no quaternion algebra, nothing shared with oracle/orc_imu.h or csrc (k_imu_eval / k_imu_blocks).

  bias correction (first order, about the bias the measurements were integrated with):
      dR_c = dR Exp(J_R_bg dbg),  dp_c = dp + J_p_ba dba + J_p_bg dbg,  dv_c = dv + J_v_ba dba + J_v_bg dbg,   [dba; dbg] = b_i - bhat
  prediction   R_pred = R_i dR_c,  p_pred = p_i + v_i dt + g dt^2 / 2 + R_i dp_c,  v_pred = v_i + g dt + R_i dv_c
  residual     [ Log(R_j^T R_pred) ; R_j^T (p_pred - p_j) ; R_j^T (v_pred - v_j) ; b_i - b_j ]                       (15)
  retractions  Pose3: X Expmap([omega; v]) (full exponential chart);  velocity, bias: vector addition
gtsam/test_vro_imu_graph.cpp:191-198, gtsam/test_ba_imu_graph.cpp:239-244, gtsam/imu_base.cpp:258-263 (gravity)."""
import mpmath as mp
import numpy as np

from tests.pose3_independent import pose_mat, expmap

mp.mp.dps = 40


def _so3_exp(w):
    S = mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return mp.expm(S)


def _so3_log(R):
    L = mp.logm(R)
    return [mp.re(L[2, 1]), mp.re(L[0, 2]), mp.re(L[1, 0])]


def _m3(a):
    return mp.matrix([[mp.mpf(float(a[3 * r + c])) for c in range(3)] for r in range(3)])


def _v(a):
    return mp.matrix([mp.mpf(float(x)) for x in a])


def residual(xi, vi, xj, vj, bi, bj, pim, g, d=None):
    """pim: tests.orc_binding.Preint (only its stored numbers are read); d: dict of tangent increments per variable name"""
    d = d or {}
    Xi, Xj = pose_mat(xi), pose_mat(xj)
    if "xi" in d: Xi = Xi * expmap(d["xi"])
    if "xj" in d: Xj = Xj * expmap(d["xj"])
    vi_, vj_, bi_, bj_ = _v(vi), _v(vj), _v(bi), _v(bj)
    if "vi" in d: vi_ = vi_ + mp.matrix(d["vi"])
    if "vj" in d: vj_ = vj_ + mp.matrix(d["vj"])
    if "bi" in d: bi_ = bi_ + mp.matrix(d["bi"])
    if "bj" in d: bj_ = bj_ + mp.matrix(d["bj"])
    Ri, Rj = Xi[0:3, 0:3], Xj[0:3, 0:3]
    pi, pj = Xi[0:3, 3], Xj[0:3, 3]
    bhat = _v(pim.bhat)
    dba, dbg = (bi_ - bhat)[0:3, 0], (bi_ - bhat)[3:6, 0]
    q = pim.dR
    dR = pose_mat([0, 0, 0, q[0], q[1], q[2], q[3]])[0:3, 0:3]
    JRbg, Jpba, Jpbg, Jvba, Jvbg = (_m3(np.asarray(m).ravel()) for m in (pim.J_R_bg, pim.J_p_ba, pim.J_p_bg, pim.J_v_ba, pim.J_v_bg))
    w = JRbg * dbg
    dRc = dR * _so3_exp([w[0], w[1], w[2]])
    dpc = _v(pim.dp) + Jpba * dba + Jpbg * dbg
    dvc = _v(pim.dv) + Jvba * dba + Jvbg * dbg
    dt = mp.mpf(float(pim.dt)); gv = _v(g)
    Rp = Ri * dRc
    pp = pi + vi_ * dt + gv * (dt * dt / 2) + Ri * dpc
    vp = vi_ + gv * dt + Ri * dvc
    rR = _so3_log(Rj.T * Rp)
    rp = Rj.T * (pp - pj)
    rv = Rj.T * (vp - vj_)
    rb = bi_ - bj_
    return list(rR) + [rp[k] for k in range(3)] + [rv[k] for k in range(3)] + [rb[k] for k in range(6)]


DIMS = {"xi": 6, "vi": 3, "xj": 6, "vj": 3, "bi": 6, "bj": 6}


def factor(xi, vi, xj, vj, bi, bj, pim, g, h=mp.mpf("1e-12")):
    """r (15) and the six Jacobians (15 x 6 / 3) in the order xi, vi, xj, vj, bi, bj"""
    r = np.array([float(x) for x in residual(xi, vi, xj, vj, bi, bj, pim, g)])
    Js = []
    for name in ("xi", "vi", "xj", "vj", "bi", "bj"):
        n = DIMS[name]
        J = np.zeros((15, n))
        for k in range(n):
            dp = [mp.mpf(0)] * n; dm = [mp.mpf(0)] * n
            dp[k] = h; dm[k] = -h
            fp = residual(xi, vi, xj, vj, bi, bj, pim, g, {name: dp})
            fm = residual(xi, vi, xj, vj, bi, bj, pim, g, {name: dm})
            J[:, k] = [float((a - b) / (2 * h)) for a, b in zip(fp, fm)]
        Js.append(J)
    return r, Js
