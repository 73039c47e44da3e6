"""graph_slam_amd — MI355X-native batch factor-graph optimiser behind the graph_slam C++ surface.

The product is the C-ABI shared library ``libfgo.so`` (``include/fgo.h``): hand-written HIP kernels for
gfx950 + a C++ host.  This Python module is only a ctypes binding used by the tests and ``bench.py``.
There is NO CPU fallback: if the library is missing this import raises, and if no HIP device is
present ``Graph()`` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FGO_LIB", os.path.join(_HERE, "libfgo.so"))      # (FGO_LIB: developer A/B of two builds on the same GPU box)

FGO_TANGENT_G2O = 0
FGO_TANGENT_GTSAM = 1


class FgoConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("verbose", C.c_int), ("ordering", C.c_int), ("nd_leaf", C.c_int), ("order_candidates", C.c_int),
                ("reserved", C.c_int * 11)]


class FgoStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("trials", C.c_int), ("terminated", C.c_int), ("structure_rebuilt", C.c_int),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("t_symbolic", C.c_double), ("t_upload", C.c_double), ("t_total", C.c_double),
                ("ms_linearize", C.c_double), ("ms_factor", C.c_double), ("ms_solve", C.c_double), ("ms_update", C.c_double),
                ("n_free", C.c_int64), ("n_edges", C.c_int64), ("nnz_H_blocks", C.c_int64), ("nnz_L_blocks", C.c_int64),
                ("n_update_ops", C.c_int64), ("n_levels", C.c_int), ("n_tasks", C.c_int),
                ("bytes_factor", C.c_double), ("bytes_linearize", C.c_double), ("bytes_solve", C.c_double),
                ("reserved", C.c_double * 8)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "reserved" else v
        return d


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)     # fgo_allreduce_fn


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "graph_slam_amd: %s is missing — build it with `python -m graph_slam_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    dp, ip, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)
    lib.fgo_create.restype = C.c_void_p
    lib.fgo_create.argtypes = [C.POINTER(FgoConfig)]
    lib.fgo_destroy.argtypes = [C.c_void_p]
    lib.fgo_last_error.restype = C.c_char_p
    lib.fgo_last_error.argtypes = [C.c_void_p]
    lib.fgo_version.restype = C.c_char_p
    lib.fgo_device_count.restype = C.c_int
    lib.fgo_add_pose.argtypes = [C.c_void_p, C.c_int64, dp, dp, C.c_int]
    lib.fgo_set_pose.argtypes = [C.c_void_p, C.c_int64, dp, dp]
    lib.fgo_get_pose.argtypes = [C.c_void_p, C.c_int64, dp]
    lib.fgo_has_pose.argtypes = [C.c_void_p, C.c_int64]
    lib.fgo_num_poses.restype = C.c_int64
    lib.fgo_num_poses.argtypes = [C.c_void_p]
    lib.fgo_num_edges.restype = C.c_int64
    lib.fgo_num_edges.argtypes = [C.c_void_p]
    lib.fgo_add_poses.argtypes = [C.c_void_p, C.c_int64, i64p, dp, C.POINTER(C.c_ubyte)]
    lib.fgo_get_poses.argtypes = [C.c_void_p, C.c_int64, i64p, dp]
    lib.fgo_add_edge_se3.argtypes = [C.c_void_p, C.c_int64, C.c_int64, dp, dp, dp, C.c_int]
    lib.fgo_add_edges_se3.argtypes = [C.c_void_p, C.c_int64, i64p, i64p, dp, dp, C.c_int]
    lib.fgo_optimize.argtypes = [C.c_void_p, C.c_int, C.POINTER(FgoStats)]
    lib.fgo_chi2.restype = C.c_double
    lib.fgo_chi2.argtypes = [C.c_void_p]
    lib.fgo_trace.argtypes = [C.c_void_p, dp, dp, C.c_int]
    lib.fgo_linearize.argtypes = [C.c_void_p, dp, dp, dp, i64p]
    lib.fgo_solve_step.argtypes = [C.c_void_p, C.c_double, dp]
    lib.fgo_bench_phase.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
    lib.fgo_get_stats.argtypes = [C.c_void_p, C.POINTER(FgoStats)]
    lib.fgo_synth_manhattan3d.restype = C.c_int64
    lib.fgo_synth_manhattan3d.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double,
                                          dp, dp, i64p, i64p, dp, dp, C.c_int64]
    lib.fgo_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, i64p, i64p]
    lib.fgo_add_prior_pose.argtypes = [C.c_void_p, C.c_int64, dp, dp, dp]
    lib.fgo_optimize_gtsam.argtypes = [C.c_void_p, C.c_int, C.POINTER(FgoStats)]
    lib.fgo_error.restype = C.c_double
    lib.fgo_isam2_update.argtypes = [C.c_void_p, C.c_double, C.POINTER(FgoStats)]
    lib.fgo_isam2_reset.argtypes = [C.c_void_p]
    lib.fgo_isam2_get_state.argtypes = [C.c_void_p, C.c_int64, dp, dp]
    lib.fgo_error.argtypes = [C.c_void_p]
    lib.fgo_marginal_cov.argtypes = [C.c_void_p, C.c_int64, dp]
    lib.fgo_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fgo_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    lib.fgo_isam2_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fgo_set_growth.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fgo_isam2_set_wildfire.argtypes = [C.c_void_p, C.c_double]
    lib.fgo_marginal_cov_many.argtypes = [C.c_void_p, C.c_int64, i64p, dp]
    lib.fgo_dist_unique_id.argtypes = [C.c_void_p]
    lib.fgo_dist_init_rccl.argtypes = [C.c_void_p, C.c_void_p]
    lib.fgo_debug_partition.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    lib.fgo_debug_allreduce.argtypes = [C.c_void_p, dp, C.c_int64]
    lib.fgo_debug_read_system.argtypes = [C.c_void_p, dp, dp, dp]
    lib.fgo_debug_read_reduced.argtypes = [C.c_void_p, C.c_double, dp, dp, i64p]
    lib.fgo_imu_params_vn100.argtypes = [dp]
    lib.fgo_preint_reset.argtypes = [dp, dp]
    lib.fgo_preint_integrate.argtypes = [dp, dp, dp, dp, C.c_double]
    lib.fgo_preint_predict.argtypes = [dp] * 7
    lib.fgo_set_fixed.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    lib.fgo_preint_information.argtypes = [dp, dp]
    lib.fgo_preint_batch.argtypes = [C.c_int, C.c_int64, C.POINTER(C.c_int64), dp, dp, C.c_double, dp, dp, dp]
    lib.fgo_add_vec3.argtypes = [C.c_void_p, C.c_int64, dp]
    lib.fgo_add_bias.argtypes = [C.c_void_p, C.c_int64, dp]
    lib.fgo_add_prior_vec3.argtypes = [C.c_void_p, C.c_int64, dp, C.c_double]
    lib.fgo_add_prior_bias.argtypes = [C.c_void_p, C.c_int64, dp, C.c_double]
    lib.fgo_set_gravity.argtypes = [C.c_void_p, dp]
    lib.fgo_add_imu_combined.argtypes = [C.c_void_p, i64p, dp]
    lib.fgo_add_plane.argtypes = [C.c_void_p, C.c_int64, dp]
    lib.fgo_add_plane_factor.argtypes = [C.c_void_p, C.c_int64, C.c_int64, dp, dp]
    lib.fgo_add_point3.argtypes = [C.c_void_p, C.c_int64, dp]
    lib.fgo_add_prior_point3.argtypes = [C.c_void_p, C.c_int64, dp, C.c_double]
    lib.fgo_set_calib_ds2.argtypes = [C.c_void_p] + [C.c_double] * 9 + [dp]
    lib.fgo_add_reproj.argtypes = [C.c_void_p, C.c_int64, C.c_int64, dp, C.c_double]
    lib.fgo_add_points3.argtypes = [C.c_void_p, C.c_int64, i64p, dp, C.c_double]
    lib.fgo_add_reprojs.argtypes = [C.c_void_p, C.c_int64, i64p, i64p, dp, C.c_double]
    return lib


lib = _load()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class FgoError(RuntimeError):
    pass


def preint_information(buf):
    """fgo_preint_information: the 15x15 information matrix fgo_add_imu_combined derives from a preintegration payload"""
    b = np.ascontiguousarray(buf, np.float64)
    out = np.zeros((15, 15))
    rc = lib.fgo_preint_information(_dp(b), _dp(out))
    if rc < 0:
        raise FgoError("fgo_preint_information failed: %d" % rc)
    return out


def preint_batch(sample_ptr, acc, gyro, dt, bias_hat=None, params=None, device=0):
    """fgo_preint_batch: every factor's samples integrated by one wave on the GPU; returns [n, PREINT_DOUBLES]"""
    sp = np.ascontiguousarray(sample_ptr, np.int64)
    n = len(sp) - 1
    a = np.ascontiguousarray(acc, np.float64); w = np.ascontiguousarray(gyro, np.float64)
    if params is None:
        params = np.zeros(IMU_PARAM_DOUBLES); lib.fgo_imu_params_vn100(_dp(params))
    params = np.ascontiguousarray(params, np.float64)
    out = np.zeros((n, PREINT_DOUBLES))
    bh = None if bias_hat is None else np.ascontiguousarray(bias_hat, np.float64)
    rc = lib.fgo_preint_batch(device, n, _i64p(sp), _dp(a), _dp(w), dt, None if bh is None else _dp(bh), _dp(params), _dp(out))
    if rc < 0:
        raise FgoError("fgo_preint_batch failed: %d" % rc)
    return out


def synth_manhattan3d(n_poses, lookback=5, n_loop=4, seed=42, sigma_t=0.02, sigma_q=0.005):
    """SURVEY.md §8d generator (host-only).  Returns dict of numpy arrays."""
    max_e = int(n_poses) * (1 + lookback + n_loop)
    init = np.zeros((n_poses, 7)); truth = np.zeros((n_poses, 7))
    ei = np.zeros(max_e, np.int64); ej = np.zeros(max_e, np.int64)
    meas = np.zeros((max_e, 7)); info = np.zeros((max_e, 21))
    e = lib.fgo_synth_manhattan3d(n_poses, lookback, n_loop, seed, sigma_t, sigma_q, _dp(init), _dp(truth),
                                  _i64p(ei), _i64p(ej), _dp(meas), _dp(info), max_e)
    if e < 0:
        raise FgoError("fgo_synth_manhattan3d failed: %d" % e)
    return dict(poses=init, truth=truth, ei=ei[:e].copy(), ej=ej[:e].copy(), meas=meas[:e].copy(), info=info[:e].copy())


def dist_unique_id():
    """ncclGetUniqueId through libfgo's run-time RCCL binding: 128 bytes rank 0 hands to every rank (Graph.init_rccl)"""
    buf = (C.c_char * 128)()
    rc = lib.fgo_dist_unique_id(C.cast(buf, C.c_void_p))
    if rc < 0:
        raise FgoError("fgo_dist_unique_id failed: %d (librccl not loadable?)" % rc)
    return bytes(buf)


def debug_partition(n, a, b, world):
    """host-only: group (owning rank, or `world` for the top) of every vertex of a block graph under fgo_set_shard(., world)"""
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    out = np.zeros(n, np.int32)
    ip = C.POINTER(C.c_int)
    rc = lib.fgo_debug_partition(n, len(a), a.ctypes.data_as(ip), b.ctypes.data_as(ip), world, out.ctypes.data_as(ip))
    if rc < 0:
        raise FgoError("fgo_debug_partition failed: %d" % rc)
    return out


class _DevArray:
    """exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n),), "typestr": "<f8", "version": 2}


def device_tensor(ptr, n, device=0):
    """torch.float64 view of `n` doubles at device address `ptr` (plumbing for the multi-GPU all-reduce hook)"""
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", device))


def torch_allreduce_hook(device=0):
    """all-reduce hook for Graph.set_shard backed by torch.distributed (backend "nccl" is RCCL on ROCm)"""
    import torch
    import torch.distributed as dist

    def hook(ptr, n):
        t = device_tensor(ptr, n, device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(device)
        return 0
    return hook


PREINT_DOUBLES = 287        # fgo_preint: dt, dR[4], dp[3], dv[3], 5 x 3x3 bias Jacobians, bhat[6], cov[225]
IMU_PARAM_DOUBLES = 9        # fgo_imu_params: 6 variances + gravity[3]


class Preintegrator:
    """Host-side mirror of the reference's imu_interface (fgo_preint_*): integrates (acc, gyro, dt) samples"""

    def __init__(self, bias_hat=None, params=None):
        self.params = np.zeros(IMU_PARAM_DOUBLES)
        lib.fgo_imu_params_vn100(_dp(self.params))
        if params is not None:
            self.params[:] = params
        self.buf = np.zeros(PREINT_DOUBLES)
        self.reset(np.zeros(6) if bias_hat is None else bias_hat)

    def reset(self, bias_hat):
        b = np.ascontiguousarray(bias_hat, np.float64)
        lib.fgo_preint_reset(_dp(self.buf), _dp(b))

    def integrate(self, acc, gyro, dt):
        a = np.ascontiguousarray(acc, np.float64); w = np.ascontiguousarray(gyro, np.float64)
        lib.fgo_preint_integrate(_dp(self.buf), _dp(self.params), _dp(a), _dp(w), dt)

    @property
    def gravity(self):
        return self.params[6:9]

    def predict(self, pose_i, vel_i, bias_i):
        xi, vi, bi = (np.ascontiguousarray(a, np.float64) for a in (pose_i, vel_i, bias_i))
        xj = np.zeros(7); vj = np.zeros(3); g = np.ascontiguousarray(self.gravity)
        lib.fgo_preint_predict(_dp(self.buf), _dp(g), _dp(xi), _dp(vi), _dp(bi), _dp(xj), _dp(vj))
        return xj, vj


class Graph:
    """Thin RAII wrapper over fgo_ctx (one per thread, like the reference's wrappers)."""

    def __init__(self, device=0, verbose=0, nd_leaf=0, order_candidates=0):
        cfg = FgoConfig()
        cfg.device, cfg.verbose, cfg.nd_leaf, cfg.order_candidates = device, verbose, nd_leaf, order_candidates
        self._h = lib.fgo_create(C.byref(cfg))
        if not self._h:
            raise FgoError("fgo_create failed: %s" % lib.fgo_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            if lib is not None:                  # (module globals are already cleared at interpreter shutdown)
                lib.fgo_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise FgoError("fgo error %d: %s" % (rc, lib.fgo_last_error(self._h).decode()))
        return rc

    def add_poses(self, poses7, fixed=None, ids=None):
        poses7 = np.ascontiguousarray(poses7, np.float64)
        n = poses7.shape[0]
        fx = None if fixed is None else np.ascontiguousarray(fixed, np.uint8)
        idp = None if ids is None else _i64p(np.ascontiguousarray(ids, np.int64))
        self._chk(lib.fgo_add_poses(self._h, n, idp, _dp(poses7),
                                    None if fx is None else fx.ctypes.data_as(C.POINTER(C.c_ubyte))))

    def add_edges(self, ei, ej, meas7, info21, tangent_order=FGO_TANGENT_G2O):
        ei = np.ascontiguousarray(ei, np.int64); ej = np.ascontiguousarray(ej, np.int64)
        meas7 = np.ascontiguousarray(meas7, np.float64); info21 = np.ascontiguousarray(info21, np.float64)
        self._chk(lib.fgo_add_edges_se3(self._h, len(ei), _i64p(ei), _i64p(ej), _dp(meas7), _dp(info21), tangent_order))

    def set_pose(self, pid, pose7):
        p = np.ascontiguousarray(pose7, np.float64)
        self._chk(lib.fgo_set_pose(self._h, pid, _dp(p[:3].copy()), _dp(p[3:].copy())))

    def get_poses(self, n=None, ids=None):
        if ids is not None:
            ids = np.ascontiguousarray(ids, np.int64); n = len(ids)
        elif n is None:
            n = lib.fgo_num_poses(self._h)
        out = np.zeros((n, 7))
        self._chk(lib.fgo_get_poses(self._h, n, None if ids is None else _i64p(ids), _dp(out)))
        return out

    def chi2(self):
        v = lib.fgo_chi2(self._h)
        if v != v:
            raise FgoError("fgo_chi2 failed: %s" % lib.fgo_last_error(self._h).decode())
        return v

    def optimize(self, iters):
        st = FgoStats()
        rc = self._chk(lib.fgo_optimize(self._h, iters, C.byref(st)))
        return rc, st

    # ---- GTSAM-semantics graph
    def add_prior(self, pid, pose7, info21):
        p = np.ascontiguousarray(pose7, np.float64); w = np.ascontiguousarray(info21, np.float64)
        self._chk(lib.fgo_add_prior_pose(self._h, pid, _dp(p[:3].copy()), _dp(p[3:].copy()), _dp(w)))

    # ---- multi-GPU shard mode
    def set_shard(self, rank, world, allreduce=None):
        """allreduce(ptr: int, count: int) -> 0 must sum `count` doubles at device address `ptr` over all ranks in place"""
        self._chk(lib.fgo_set_shard(self._h, rank, world))
        self.rank, self.world = rank, world
        if allreduce is not None:
            self._ar_cb = ALLREDUCE_FN(lambda user, ptr, n: int(allreduce(ptr, n) or 0))   # keep a reference alive
            self._chk(lib.fgo_set_allreduce(self._h, self._ar_cb, None))

    def init_rccl(self, id128):
        """RCCL transport for the collectives of the distributed mode (after set_shard); id128: bytes from dist_unique_id()"""
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._chk(lib.fgo_dist_init_rccl(self._h, C.cast(buf, C.c_void_p)))

    def debug_allreduce(self, a):
        a = np.ascontiguousarray(a, np.float64).copy()
        self._chk(lib.fgo_debug_allreduce(self._h, _dp(a), a.size))
        return a

    def read_system(self):
        st = FgoStats()
        self._chk(lib.fgo_debug_read_system(self._h, None, None, None))              # builds the structure (no collective)
        self._chk(lib.fgo_get_stats(self._h, C.byref(st)))
        H = np.zeros(int(st.nnz_H_blocks) * 36); b = np.zeros(int(st.n_free) * 6); chi = C.c_double()
        self._chk(lib.fgo_debug_read_system(self._h, _dp(H), _dp(b), C.byref(chi)))
        return H, b, chi.value

    def read_reduced(self, lam=0.0):
        """dense reduced camera system (S, g) of a structure with the landmarks eliminated (fgo_debug_read_reduced)"""
        n = C.c_int64()
        self._chk(lib.fgo_debug_read_reduced(self._h, lam, None, None, C.byref(n)))
        m = 6 * int(n.value)
        S, g = np.zeros((m, m)), np.zeros(m)
        self._chk(lib.fgo_debug_read_reduced(self._h, lam, _dp(S), _dp(g), C.byref(n)))
        return S, g

    def add_vec3(self, pid, xyz):
        self._chk(lib.fgo_add_vec3(self._h, pid, _dp(np.ascontiguousarray(xyz, np.float64))))

    def add_bias(self, pid, b6):
        self._chk(lib.fgo_add_bias(self._h, pid, _dp(np.ascontiguousarray(b6, np.float64))))

    def add_prior_vec3(self, pid, xyz, sigma):
        self._chk(lib.fgo_add_prior_vec3(self._h, pid, _dp(np.ascontiguousarray(xyz, np.float64)), sigma))

    def add_prior_bias(self, pid, b6, sigma):
        self._chk(lib.fgo_add_prior_bias(self._h, pid, _dp(np.ascontiguousarray(b6, np.float64)), sigma))

    def set_gravity(self, g3):
        self._chk(lib.fgo_set_gravity(self._h, _dp(np.ascontiguousarray(g3, np.float64))))

    def add_imu(self, ids6, preint_buf):
        ids = np.ascontiguousarray(ids6, np.int64); buf = np.ascontiguousarray(preint_buf, np.float64)
        assert len(buf) == PREINT_DOUBLES
        self._chk(lib.fgo_add_imu_combined(self._h, _i64p(ids), _dp(buf)))

    def marginal_cov(self, pid):
        out = np.zeros((6, 6))
        self._chk(lib.fgo_marginal_cov(self._h, pid, _dp(out)))
        return out

    def marginal_cov_many(self, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.zeros((len(ids), 6, 6))
        self._chk(lib.fgo_marginal_cov_many(self._h, len(ids), _i64p(ids), _dp(out)))
        return out

    def add_plane(self, pid, abcd):
        a = np.ascontiguousarray(abcd, np.float64)
        self._chk(lib.fgo_add_plane(self._h, pid, _dp(a)))

    def add_plane_factor(self, pose_id, plane_id, z_abcd, cov_ut6):
        z = np.ascontiguousarray(z_abcd, np.float64); s = np.ascontiguousarray(cov_ut6, np.float64)
        self._chk(lib.fgo_add_plane_factor(self._h, pose_id, plane_id, _dp(z), _dp(s)))

    def add_point(self, pid, xyz):
        a = np.ascontiguousarray(xyz, np.float64)
        self._chk(lib.fgo_add_point3(self._h, pid, _dp(a)))

    def add_prior_point(self, pid, xyz, sigma):
        a = np.ascontiguousarray(xyz, np.float64)
        self._chk(lib.fgo_add_prior_point3(self._h, pid, _dp(a), sigma))

    def set_calibration(self, calib9, body_P_sensor7=None):
        c9 = [float(x) for x in calib9]
        b = None if body_P_sensor7 is None else _dp(np.ascontiguousarray(body_P_sensor7, np.float64))
        self._chk(lib.fgo_set_calib_ds2(self._h, *c9, b))

    def add_reproj(self, pose_id, point_id, uv, sigma=1.0):
        a = np.ascontiguousarray(uv, np.float64)
        self._chk(lib.fgo_add_reproj(self._h, pose_id, point_id, _dp(a), sigma))

    def optimize_gtsam(self, max_iters=100):
        st = FgoStats()
        rc = self._chk(lib.fgo_optimize_gtsam(self._h, max_iters, C.byref(st)))
        return rc, st

    def isam2_update(self, relinearize_threshold=0.1):
        """ISAM2::update(new factors, new values) + calculateEstimate() (gtsam_graph.cpp:1768-1776)"""
        st = FgoStats()
        self._chk(lib.fgo_isam2_update(self._h, relinearize_threshold, C.byref(st)))
        return st

    def isam2_set_wildfire(self, threshold):
        """ISAM2Params::wildfireThreshold analogue; 0 (default) = exact back-substitution"""
        self._chk(lib.fgo_isam2_set_wildfire(self._h, threshold))

    def isam2_reserve(self, reserve_variables, window=0):
        self._chk(lib.fgo_isam2_reserve(self._h, reserve_variables, window))

    def set_growth(self, reserve_variables, window=0):
        """growth reserve of a g2o-semantics context (fgo_set_growth): new vertices / local edges without a structure rebuild"""
        self._chk(lib.fgo_set_growth(self._h, reserve_variables, window))

    def isam2_reset(self):
        self._chk(lib.fgo_isam2_reset(self._h))

    def isam2_state(self, pid):
        th = np.zeros(7); de = np.zeros(6)
        self._chk(lib.fgo_isam2_get_state(self._h, pid, _dp(th), _dp(de)))
        return th, de

    def error(self):
        v = lib.fgo_error(self._h)
        if v != v:
            raise FgoError("fgo_error failed: %s" % lib.fgo_last_error(self._h).decode())
        return v

    def trace(self, cap=256):
        a = np.zeros(cap); b = np.zeros(cap)
        m = self._chk(lib.fgo_trace(self._h, _dp(a), _dp(b), cap))
        return a[:m], b[:m]

    def linearize(self, dense=True):
        chi = C.c_double(); nf = C.c_int64()
        if not dense:
            self._chk(lib.fgo_linearize(self._h, C.byref(chi), None, None, C.byref(nf)))
            return chi.value, None, None
        self._chk(lib.fgo_linearize(self._h, C.byref(chi), None, None, C.byref(nf)))
        m = 6 * nf.value
        H = np.zeros((m, m)); b = np.zeros(m)
        self._chk(lib.fgo_linearize(self._h, C.byref(chi), _dp(H), _dp(b), C.byref(nf)))
        return chi.value, H, b

    def solve_step(self, lam):
        chi = C.c_double(); nf = C.c_int64()
        self._chk(lib.fgo_linearize(self._h, C.byref(chi), None, None, C.byref(nf)))
        d = np.zeros(6 * nf.value)
        self._chk(lib.fgo_solve_step(self._h, lam, _dp(d)))
        return d

    def bench_phase(self, phase, reps):
        ms = C.c_double()
        self._chk(lib.fgo_bench_phase(self._h, phase, reps, C.byref(ms)))
        return ms.value

    def stats(self):
        st = FgoStats()
        self._chk(lib.fgo_get_stats(self._h, C.byref(st)))
        return st
