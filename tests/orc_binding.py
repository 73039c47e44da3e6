"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py — never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "liborc.so")


class OrcStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("trials", C.c_int), ("terminated", C.c_int),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("t_symbolic", C.c_double), ("t_linearize", C.c_double), ("t_factor", C.c_double),
                ("t_solve", C.c_double), ("t_update", C.c_double), ("t_total", C.c_double),
                ("nnz_H_blocks", C.c_longlong), ("nnz_L_scalar", C.c_longlong)]


def _load(path=LIB):
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    if path == LIB and (not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)):
        subprocess.run(["make", "-s", "-C", ODIR], check=True)
    lib = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.orc_edge_se3_eval.argtypes = [dp] * 6
    lib.orc_pose_oplus_eval.argtypes = [dp] * 3
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_int, dp, C.POINTER(C.c_ubyte), C.c_int, ip, ip, dp, dp]
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_chi2.restype = C.c_double
    lib.orc_chi2.argtypes = [C.c_void_p]
    lib.orc_optimize.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcStats)]
    lib.orc_get_poses.argtypes = [C.c_void_p, dp]
    lib.orc_set_poses.argtypes = [C.c_void_p, dp]
    lib.orc_dense_system.argtypes = [C.c_void_p, dp, dp, ip]
    lib.orc_trace.argtypes = [C.c_void_p, dp, dp, C.c_int]
    lib.orc_solve_step.argtypes = [C.c_void_p, C.c_double, dp]
    lib.orc_amd_order.argtypes = [C.c_int, ip, ip, ip]
    # GTSAM-semantics extension
    lib.orc_set_gtsam.argtypes = [C.c_void_p]
    lib.orc_add_priors.argtypes = [C.c_void_p, C.c_int, ip, dp, dp]
    lib.orc_error_gtsam.restype = C.c_double
    lib.orc_error_gtsam.argtypes = [C.c_void_p]
    lib.orc_optimize_gtsam.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcStats)]
    lib.orc_isam2_step.argtypes = [C.c_void_p, C.c_double, dp, dp, dp, C.POINTER(C.c_int)]
    lib.orc_between_eval.argtypes = [dp] * 6
    lib.orc_prior_eval.argtypes = [dp] * 4
    lib.orc_pose3_retract_eval.argtypes = [dp] * 3
    lib.orc_pose3_logmap_eval.argtypes = [dp] * 2
    lib.orc_pose3_expmap_eval.argtypes = [dp] * 2
    lib.orc_plane_make.argtypes = [dp] * 2
    lib.orc_plane_transform_eval.argtypes = [dp] * 5
    lib.orc_plane_retract_eval.argtypes = [dp] * 3
    lib.orc_plane_local_eval.argtypes = [dp] * 3
    lib.orc_plane_error_vector_eval.argtypes = [dp] * 3
    lib.orc_plane_factor_eval.argtypes = [dp] * 6
    lib.orc_reproj_eval.argtypes = [dp] * 8
    lib.orc_preint_size.restype = C.c_int
    lib.orc_preint_run.argtypes = [C.c_void_p, dp, C.c_int, dp, dp, C.c_double]
    lib.orc_preint_run_params.argtypes = [C.c_void_p, dp, C.c_int, dp, dp, C.c_double, dp]
    lib.orc_imu_factor_eval.argtypes = [dp] * 6 + [C.c_void_p] + [dp] * 8
    lib.orc_imu_predict.argtypes = [dp] * 3 + [C.c_void_p] + [dp] * 3
    lib.orc_add_imu_factors.argtypes = [C.c_void_p, C.c_int, ip, C.c_void_p, dp, dp]
    lib.orc_set_var_kinds.argtypes = [C.c_void_p, ip]
    lib.orc_set_edge_kinds.argtypes = [C.c_void_p, ip]
    lib.orc_set_calibration.argtypes = [C.c_void_p, dp, dp]
    return lib


lib = _load()
NATIVE = None          # set by use_native(): {"march": ..., "lib": ...}


def use_native():
    """bench.py's cpu_baseline leg: rebuild the oracle ON THIS HOST with -march=native (liborc_native.so) and switch this
    binding to it, so that the CPU figure is from a binary scheduled for the cores it is timed on.  Returns a dict with the
    -march gcc resolved `native` to; falls back to the portable build (and says so) if the box has no compiler."""
    global lib, NATIVE
    info = {"lib": "liborc.so", "march": "x86-64-v3 (portable build: native rebuild failed)"}
    try:
        subprocess.run(["make", "-s", "-B", "-C", ODIR, "native"], check=True, capture_output=True, timeout=600)
        q = subprocess.run(["gcc", "-march=native", "-Q", "--help=target"], capture_output=True, text=True, timeout=60).stdout
        march = [l.split()[-1] for l in q.splitlines() if l.strip().startswith("-march=")]
        lib = _load(os.path.join(ODIR, "liborc_native.so"))
        ver = subprocess.run(["gcc", "--version"], capture_output=True, text=True, timeout=60).stdout.splitlines()
        # (the compiler is named because `native` resolves to the newest core THIS gcc knows: an 11.x gcc calls a Zen 5 part znver3)
        info = {"lib": "liborc_native.so", "march": "native = " + (march[0] if march else "?"), "flags": "-O3 -march=native -fopenmp",
                "compiler": ver[0].strip() if ver else "?"}
    except (OSError, subprocess.SubprocessError) as ex:
        info["error"] = str(ex)[:200]
    NATIVE = info
    return info


def set_threads(n):
    """threads of the oracle's OpenMP leg (linearisation + sub-tree parallel numeric Cholesky); 1 = the scalar port"""
    lib.orc_set_threads(int(n))


def set_solver(kind):
    """0 = simplicial (what the reference links), 1 = supernodal (oracle/orc_chol_sn.c)"""
    lib.orc_set_solver(int(kind))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def edge_se3(xi, xj, z, jac=True):
    xi, xj, z = (np.ascontiguousarray(a, np.float64) for a in (xi, xj, z))
    e = np.zeros(6); Ji = np.zeros((6, 6)); Jj = np.zeros((6, 6))
    lib.orc_edge_se3_eval(_dp(xi), _dp(xj), _dp(z), _dp(e), _dp(Ji) if jac else None, _dp(Jj) if jac else None)
    return (e, Ji, Jj) if jac else e


def oplus(x, d):
    x = np.ascontiguousarray(x, np.float64); d = np.ascontiguousarray(d, np.float64)
    out = np.zeros(7)
    lib.orc_pose_oplus_eval(_dp(x), _dp(d), _dp(out))
    return out


def between(xi, xj, z, jac=True):
    xi, xj, z = (np.ascontiguousarray(a, np.float64) for a in (xi, xj, z))
    e = np.zeros(6); Ji = np.zeros((6, 6)); Jj = np.zeros((6, 6))
    lib.orc_between_eval(_dp(xi), _dp(xj), _dp(z), _dp(e), _dp(Ji) if jac else None, _dp(Jj) if jac else None)
    return (e, Ji, Jj) if jac else e


def prior(x, mean, jac=True):
    x, mean = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(mean, np.float64)
    e = np.zeros(6); J = np.zeros((6, 6))
    lib.orc_prior_eval(_dp(x), _dp(mean), _dp(e), _dp(J) if jac else None)
    return (e, J) if jac else e


def retract(x, xi):
    x, xi = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(xi, np.float64)
    out = np.zeros(7); lib.orc_pose3_retract_eval(_dp(x), _dp(xi), _dp(out)); return out


def logmap(T):
    T = np.ascontiguousarray(T, np.float64); xi = np.zeros(6); lib.orc_pose3_logmap_eval(_dp(T), _dp(xi)); return xi


def expmap(xi):
    xi = np.ascontiguousarray(xi, np.float64); T = np.zeros(7); lib.orc_pose3_expmap_eval(_dp(xi), _dp(T)); return T


def plane(a, b, c, d):
    """OrientedPlane3(a, b, c, d): unit normal + distance"""
    p = np.zeros(4); lib.orc_plane_make(_dp(np.array([a, b, c, d], np.float64)), _dp(p)); return p


def plane_transform(p, x, jac=False):
    p, x = np.ascontiguousarray(p, np.float64), np.ascontiguousarray(x, np.float64)
    out = np.zeros(4); Hx = np.zeros((3, 6)); Hp = np.zeros((3, 3))
    lib.orc_plane_transform_eval(_dp(p), _dp(x), _dp(out), _dp(Hx) if jac else None, _dp(Hp) if jac else None)
    return (out, Hx, Hp) if jac else out


def plane_retract(p, v):
    p, v = np.ascontiguousarray(p, np.float64), np.ascontiguousarray(v, np.float64)
    out = np.zeros(4); lib.orc_plane_retract_eval(_dp(p), _dp(v), _dp(out)); return out


def plane_local(p, q):
    p, q = np.ascontiguousarray(p, np.float64), np.ascontiguousarray(q, np.float64)
    v = np.zeros(3); lib.orc_plane_local_eval(_dp(p), _dp(q), _dp(v)); return v


def plane_error_vector(p, o):
    p, o = np.ascontiguousarray(p, np.float64), np.ascontiguousarray(o, np.float64)
    e = np.zeros(3); lib.orc_plane_error_vector_eval(_dp(p), _dp(o), _dp(e)); return e


def plane_factor(x, pl, z, jac=True):
    x, pl, z = (np.ascontiguousarray(a, np.float64) for a in (x, pl, z))
    r = np.zeros(3); Hx = np.zeros((3, 6)); Hp = np.zeros((3, 3))
    lib.orc_plane_factor_eval(_dp(x), _dp(pl), _dp(z), _dp(r), _dp(Hx) if jac else None, _dp(Hp) if jac else None)
    return (r, Hx, Hp) if jac else r


GRAVITY = np.array([0.0, 0.0, 9.71])          # MakeSharedD(9.71): imu_base.cpp:258-263


class Preint:
    """PreintegratedCombinedMeasurements (oracle restatement): layout of struct orc_preint as float64 fields"""
    FIELDS = [("dt", 1), ("dR", 4), ("dp", 3), ("dv", 3), ("J_R_bg", 9), ("J_p_ba", 9), ("J_p_bg", 9), ("J_v_ba", 9),
              ("J_v_bg", 9), ("bhat", 6), ("cov", 225)]

    def __init__(self, bhat, acc, gyro, dt, variances=None):
        """variances: None = the reference's VN100 settings, else (acc, gyro, integration, bias acc, bias gyro, biasAccOmegaInt)"""
        n = lib.orc_preint_size() // 8
        assert n == sum(k for _, k in self.FIELDS)
        self.buf = np.zeros(n)
        acc = np.ascontiguousarray(acc, np.float64); gyro = np.ascontiguousarray(gyro, np.float64)
        bhat = np.ascontiguousarray(bhat, np.float64)
        if variances is None:
            lib.orc_preint_run(self.buf.ctypes.data, _dp(bhat), len(acc), _dp(acc), _dp(gyro), C.c_double(dt))
        else:
            v = np.ascontiguousarray(variances, np.float64)
            lib.orc_preint_run_params(self.buf.ctypes.data, _dp(bhat), len(acc), _dp(acc), _dp(gyro), C.c_double(dt), _dp(v))

    def __getattr__(self, name):
        o = 0
        for f, k in self.FIELDS:
            if f == name:
                v = self.buf[o:o + k]
                return float(v[0]) if k == 1 else (v.reshape(15, 15) if k == 225 else (v.reshape(3, 3) if k == 9 else v))
            o += k
        raise AttributeError(name)

    def predict(self, xi, vi, bi, g=GRAVITY):
        xi, vi, bi, g = (np.ascontiguousarray(a, np.float64) for a in (xi, vi, bi, g))
        xj = np.zeros(7); vj = np.zeros(3)
        lib.orc_imu_predict(_dp(xi), _dp(vi), _dp(bi), self.buf.ctypes.data, _dp(g), _dp(xj), _dp(vj))
        return xj, vj

    def factor(self, xi, vi, xj, vj, bi, bj, jac=True, g=GRAVITY):
        a = [np.ascontiguousarray(x, np.float64) for x in (xi, vi, xj, vj, bi, bj)]
        g = np.ascontiguousarray(g, np.float64)
        r = np.zeros(15)
        Js = [np.zeros((15, k)) for k in (6, 3, 6, 3, 6, 6)]
        lib.orc_imu_factor_eval(*[_dp(x) for x in a], self.buf.ctypes.data, _dp(g), _dp(r), *[(_dp(J) if jac else None) for J in Js])
        return (r, Js) if jac else r


VK_POSE, VK_PLANE, VK_POINT, VK_VEC3, VK_BIAS = 0, 1, 2, 3, 4
FK_G2O, FK_BETWEEN, FK_PLANE, FK_REPROJ = 0, 1, 2, 3


def reproj(x, pw, uv, calib, bps, jac=True):
    x, pw, uv, calib, bps = (np.ascontiguousarray(a, np.float64) for a in (x, pw, uv, calib, bps))
    r = np.zeros(2); Hx = np.zeros((2, 6)); Hp = np.zeros((2, 3))
    lib.orc_reproj_eval(_dp(x), _dp(pw), _dp(uv), _dp(calib), _dp(bps), _dp(r), _dp(Hx) if jac else None, _dp(Hp) if jac else None)
    return (r, Hx, Hp) if jac else r


class Problem:
    def set_gtsam(self):
        lib.orc_set_gtsam(self._h)

    def set_kinds(self, var_kinds, edge_kinds):
        vk = np.ascontiguousarray(var_kinds, np.int32); ek = np.ascontiguousarray(edge_kinds, np.int32)
        lib.orc_set_var_kinds(self._h, _ip(vk)); lib.orc_set_edge_kinds(self._h, _ip(ek))

    def add_imu_factors(self, ids6, preints, infos225, gravity=None):
        ids = np.ascontiguousarray(ids6, np.int32).reshape(-1, 6)
        buf = np.ascontiguousarray(np.concatenate([p.buf for p in preints]))
        info = np.ascontiguousarray(infos225, np.float64).reshape(len(ids), 225)
        g = np.ascontiguousarray(GRAVITY if gravity is None else gravity, np.float64)
        lib.orc_add_imu_factors(self._h, len(ids), _ip(ids), buf.ctypes.data, _dp(info), _dp(g))

    def set_calibration(self, calib9, bps7):
        c = np.ascontiguousarray(calib9, np.float64); b = np.ascontiguousarray(bps7, np.float64)
        lib.orc_set_calibration(self._h, _dp(c), _dp(b))

    def add_priors(self, ids, mean7, info21):
        ids = np.ascontiguousarray(ids, np.int32); mean7 = np.ascontiguousarray(mean7, np.float64)
        info21 = np.ascontiguousarray(info21, np.float64)
        lib.orc_add_priors(self._h, len(ids), _ip(ids), _dp(mean7), _dp(info21))

    def error_gtsam(self):
        return lib.orc_error_gtsam(self._h)

    def optimize_gtsam(self, max_iters=100):
        st = OrcStats()
        rc = lib.orc_optimize_gtsam(self._h, max_iters, C.byref(st))
        return rc, st

    def isam2_step(self, threshold, theta7, delta6):
        """in-place on theta7 (N x 7) / delta6 (N x 6); returns (estimate N x 7, variables relinearised)"""
        assert theta7.flags.c_contiguous and delta6.flags.c_contiguous and theta7.shape == (self.N, 7) and delta6.shape == (self.N, 6)
        est = np.zeros((self.N, 7)); n = C.c_int()
        rc = lib.orc_isam2_step(self._h, threshold, _dp(theta7), _dp(delta6), _dp(est), C.byref(n))
        if rc:
            raise RuntimeError("orc_isam2_step failed: %d" % rc)
        return est, n.value

    def __init__(self, poses, fixed, ei, ej, meas, info):
        self.poses = np.ascontiguousarray(poses, np.float64)
        self.fixed = np.ascontiguousarray(fixed, np.uint8)
        self.ei = np.ascontiguousarray(ei, np.int32); self.ej = np.ascontiguousarray(ej, np.int32)
        self.meas = np.ascontiguousarray(meas, np.float64); self.info = np.ascontiguousarray(info, np.float64)
        self.N, self.E = len(self.poses), len(self.ei)
        self._h = lib.orc_create(self.N, _dp(self.poses), self.fixed.ctypes.data_as(C.POINTER(C.c_ubyte)), self.E,
                                 _ip(self.ei), _ip(self.ej), _dp(self.meas), _dp(self.info))
        self.nfree = int((self.fixed == 0).sum())

    def __del__(self):
        if getattr(self, "_h", None):
            lib.orc_free(self._h); self._h = None

    def chi2(self):
        return lib.orc_chi2(self._h)

    def optimize(self, iters):
        st = OrcStats()
        rc = lib.orc_optimize(self._h, iters, C.byref(st))
        return rc, st

    def get_poses(self):
        out = np.zeros((self.N, 7)); lib.orc_get_poses(self._h, _dp(out)); return out

    def set_poses(self, p):
        p = np.ascontiguousarray(p, np.float64); lib.orc_set_poses(self._h, _dp(p))

    def dense_system(self):
        m = 6 * self.nfree
        H = np.zeros((m, m)); b = np.zeros(m); n = C.c_int()
        lib.orc_dense_system(self._h, _dp(H), _dp(b), C.byref(n))
        return H, b

    def trace(self, cap=256):
        a = np.zeros(cap); b = np.zeros(cap)
        m = lib.orc_trace(self._h, _dp(a), _dp(b), cap)
        return a[:m], b[:m]

    def solve_step(self, lam):
        d = np.zeros(6 * self.nfree)
        rc = lib.orc_solve_step(self._h, lam, _dp(d))
        return rc, d
