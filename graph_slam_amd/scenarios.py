"""Synthetic workloads for BASELINE.json configs 3 (bundle adjustment) and 4 (visual-inertial + planes), built through
the C-ABI exactly the way CGraphGT assembles them.  Host-only numpy + the host preintegrator; used by
the tests and by bench.py's optional workloads.  Nothing here touches the oracle.
"""
import numpy as np

from . import Graph, Preintegrator, FGO_TANGENT_GTSAM, lib, _dp, _i64p

SR4000 = (250.5773, 250.5773, 0.0, 90.0, 70.0, -0.8466, 0.5370, 0.0, 0.0)     # gtsam/test_ba_imu_graph.cpp:84


def _quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + bw * ax + ay * bz - az * by, aw * by + bw * ay + az * bx - ax * bz,
                     aw * bz + bw * az + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], -1)


def _quat_rot(q, v):
    u, w = q[..., :3], q[..., 3:4]
    c = np.cross(u, v)
    return v + 2 * (w * c + np.cross(u, c))


def _project(pc, calib):
    fx, fy, s, u0, v0, k1, k2, p1, p2 = calib
    x, y = pc[..., 0] / pc[..., 2], pc[..., 1] / pc[..., 2]
    rr = x * x + y * y
    g = 1 + k1 * rr + k2 * rr * rr
    xd = g * x + 2 * p1 * x * y + p2 * (rr + 2 * x * x)
    yd = g * y + 2 * p2 * x * y + p1 * (rr + 2 * y * y)
    return np.stack([fx * xd + s * yd + u0, fy * yd + v0], -1)


def ba_problem(n_kf=10000, n_pts=500000, obs_per_pt=10, seed=43, calib=SR4000, pixel_sigma=1.0, point_sigma=0.014):
    """config 3: keyframes on a gently weaving path, points in the slab in view, every point observed by the
    `obs_per_pt` nearest-in-time keyframes that see it (gtsam_graph.cpp:370-448: Cal3DS2, PriorFactor<Point3> sigma
    0.014, GenericProjectionFactor sigma 1 px, body_P_sensor).  Returns a dict of numpy arrays (truth + noisy start)."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_kf)
    t = np.stack([0.05 * k, 0.3 * np.sin(0.002 * k), 0.1 * np.sin(0.0031 * k)], 1)
    yaw = 0.15 * np.sin(0.0017 * k)
    q = np.stack([np.zeros(n_kf), np.zeros(n_kf), np.sin(yaw / 2), np.cos(yaw / 2)], 1)
    poses = np.concatenate([t, q], 1)
    bps = np.array([0.0, 0.0, 0.0, 0.5, 0.5, 0.5, 0.5])        # body x-forward -> camera z-forward style axis swap
    bq = np.broadcast_to(bps[3:], (n_kf, 4))
    cam_q = _quat_mul(q, bq)
    cam_t = t + _quat_rot(q, np.broadcast_to(bps[:3], (n_kf, 3)))
    centre = rng.integers(0, n_kf, n_pts)
    pc = np.stack([rng.uniform(-0.3, 0.3, n_pts), rng.uniform(-0.25, 0.25, n_pts), np.ones(n_pts)], 1) * rng.uniform(2.0, 5.0, (n_pts, 1))
    pw = cam_t[centre] + _quat_rot(cam_q[centre], pc)
    half = obs_per_pt // 2
    kf_idx = (centre[:, None] + np.arange(-half, obs_per_pt - half)[None, :]).clip(0, n_kf - 1)
    pt_idx = np.broadcast_to(np.arange(n_pts)[:, None], kf_idx.shape)
    kf_flat, pt_flat = kf_idx.reshape(-1), pt_idx.reshape(-1)
    cq = cam_q[kf_flat] * np.array([-1, -1, -1, 1.0])
    pk = _quat_rot(cq, pw[pt_flat] - cam_t[kf_flat])
    vis = (pk[:, 2] > 0.5) & (np.abs(pk[:, 0] / pk[:, 2]) < 0.45) & (np.abs(pk[:, 1] / pk[:, 2]) < 0.45)
    # drop duplicate (kf, pt) pairs created by clipping at the ends of the path
    key = kf_flat.astype(np.int64) * n_pts + pt_flat
    _, first = np.unique(key, return_index=True)
    keep = np.zeros(len(key), bool); keep[first] = True
    sel = vis & keep
    kf_flat, pt_flat, pk = kf_flat[sel], pt_flat[sel], pk[sel]
    uv = _project(pk, calib) + rng.normal(size=(len(kf_flat), 2)) * pixel_sigma
    poses0 = poses.copy()
    poses0[1:, :3] += rng.normal(size=(n_kf - 1, 3)) * 0.01
    pts0 = pw + rng.normal(size=pw.shape) * point_sigma
    return dict(poses=poses, poses0=poses0, points=pw, points0=pts0, obs_kf=kf_flat, obs_pt=pt_flat, obs_uv=uv, calib=calib,
                bps=bps, pixel_sigma=pixel_sigma, point_sigma=point_sigma)


def ba_graph(p, device=0, pose_prior_sigma=1e-3, odometry_sigma=(0.002, 0.005)):
    """assemble config 3 through the C-ABI: X(k) poses, Q(j) points (ids offset by n_kf), priors, projection factors and
    the VO BetweenFactors the offline driver adds between consecutive keyframes (gtsam_graph.cpp:1593-1623)"""
    n_kf, n_pts = len(p["poses0"]), len(p["points0"])
    gr = Graph(device=device)
    gr.add_poses(p["poses0"])
    w = np.zeros(21); idx = [0, 6, 11, 15, 18, 20]
    w[idx] = 1.0 / pose_prior_sigma ** 2
    gr.add_prior(0, p["poses"][0], w)
    pts = np.ascontiguousarray(p["points0"])
    pid = np.arange(n_kf, n_kf + n_pts, dtype=np.int64)
    gr._chk(lib.fgo_add_points3(gr._h, n_pts, _i64p(pid), _dp(pts), p["point_sigma"]))
    gr.set_calibration(p["calib"], p["bps"])
    uv = np.ascontiguousarray(p["obs_uv"])
    okf = np.ascontiguousarray(p["obs_kf"], np.int64); opt = np.ascontiguousarray(n_kf + p["obs_pt"], np.int64)
    gr._chk(lib.fgo_add_reprojs(gr._h, len(uv), _i64p(okf), _i64p(opt), _dp(uv), p["pixel_sigma"]))
    # odometry (BetweenFactor<Pose3>) between consecutive keyframes, from the truth
    a, b = p["poses"][:-1], p["poses"][1:]
    qa_c = a[:, 3:] * np.array([-1, -1, -1, 1.0])
    zt = _quat_rot(qa_c, b[:, :3] - a[:, :3]); zq = _quat_mul(qa_c, b[:, 3:])
    meas = np.concatenate([zt, zq], 1)
    wi = np.zeros(21); wi[[0, 6, 11]] = 1.0 / odometry_sigma[0] ** 2; wi[[15, 18, 20]] = 1.0 / odometry_sigma[1] ** 2
    gr.add_edges(np.arange(n_kf - 1), np.arange(1, n_kf), meas, np.tile(wi, (n_kf - 1, 1)), tangent_order=FGO_TANGENT_GTSAM)
    return gr


def vio_problem(n_kf=50000, samples=40, n_planes=200, lookback=4, seed=44, noise=0.01):
    """config 4: keyframes at 5 Hz, IMU at 200 Hz (40 samples per factor, dt = 0.005: test_vro_imu_graph.cpp:111), VN100
    noise, gravity 9.71; BetweenFactors with config-1 topology; `n_planes` plane landmarks, each keyframe observes 1-3
    (Sigma = diag(1e-4): gtsam_graph.cpp:1206); priors as CGraphGT::firstNode (gtsam_graph.cpp:338-367)."""
    rng = np.random.default_rng(seed)
    bias_true = np.array([0.03, -0.02, 0.01, 0.002, -0.001, 0.0015])
    X = np.zeros((n_kf, 7)); X[0, 6] = 1.0
    V = np.zeros((n_kf, 3)); V[0] = [0.3, 0.1, 0.0]
    pre = np.zeros((n_kf - 1, 287))
    pim = Preintegrator()
    tt = np.arange(samples) * 0.005
    for k in range(n_kf - 1):
        ph = rng.uniform(0, 6, 6)
        gyro = 0.25 * np.sin(2.1 * tt[:, None] + ph[None, :3]) + bias_true[3:]
        # specific force: cancels gravity on average so the platform stays bounded
        acc = 0.6 * np.cos(1.3 * tt[:, None] + ph[None, 3:]) + bias_true[:3]
        Rw = X[k, 3:]
        g_body = _quat_rot(Rw * np.array([-1, -1, -1, 1.0]), np.array([0, 0, -9.71]))
        acc = acc + g_body[None, :] - 0.5 * _quat_rot(Rw * np.array([-1, -1, -1, 1.0]), V[k])[None, :]   # damp the velocity
        pim.reset(np.zeros(6))
        for a, w in zip(acc, gyro):
            pim.integrate(a, w, 0.005)
        X[k + 1], V[k + 1] = pim.predict(X[k], V[k], bias_true)
        pre[k] = pim.buf
    planes = np.zeros((n_planes, 4))
    nrm = rng.normal(size=(n_planes, 3)); planes[:, :3] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    planes[:, 3] = rng.uniform(2.0, 8.0, n_planes)
    return dict(X=X, V=V, bias=bias_true, pre=pre, planes=planes, lookback=lookback, noise=noise, seed=seed, gravity=pim.gravity.copy())


def vio_factors(p):
    """everything config 4's graph holds besides the IMU payloads, as plain arrays (the same random stream vio_graph always
    used): start values, between factors, plane observations.  ids: X(k) = k, V(k) = K + k, B(k) = 2K + k, L(j) = 3K + j"""
    rng = np.random.default_rng(p["seed"] + 1)
    X, V, K = p["X"], p["V"], len(p["X"])
    nz = p["noise"]
    X0 = X.copy(); X0[1:, :3] += rng.normal(size=(K - 1, 3)) * 0.03
    V0 = V + rng.normal(size=V.shape) * 0.05; V0[0] = V[0]
    planes0 = p["planes"].copy()
    for j in range(len(planes0)):
        planes0[j, 3] += rng.normal() * 0.05
    # between factors: odometry + look-back, measurements from the truth + noise
    ei, ej = [], []
    for d in range(1, p["lookback"] + 2):
        ei.append(np.arange(0, K - d)); ej.append(np.arange(d, K))
    ei, ej = np.concatenate(ei), np.concatenate(ej)
    qa_c = X[ei, 3:] * np.array([-1, -1, -1, 1.0])
    zt = _quat_rot(qa_c, X[ej, :3] - X[ei, :3]) + rng.normal(size=(len(ei), 3)) * nz
    dq = np.concatenate([rng.normal(size=(len(ei), 3)) * nz * 0.5, np.ones((len(ei), 1))], 1)
    zq = _quat_mul(_quat_mul(qa_c, X[ej, 3:]), dq / np.linalg.norm(dq, axis=1, keepdims=True))
    wi = np.zeros(21); wi[[0, 6, 11]] = 1.0 / nz ** 2; wi[[15, 18, 20]] = 1.0 / (2 * nz) ** 2
    npl = len(p["planes"])
    # planes 0 and 1 play floor / ceiling (seen from everywhere); the others are walls seen only while the platform
    # is near them: wall j is visible from keyframes within +-span of its centre keyframe
    centres = np.linspace(0, K - 1, max(npl - 2, 1))
    span = max(8.0, 1.5 * K / max(npl - 2, 1))
    pl_kf, pl_id, pl_z = [], [], []
    for k in range(K):
        near = 2 + np.nonzero(np.abs(centres - k) <= span)[0] if npl > 2 else np.zeros(0, int)
        cand = np.concatenate([[0, 1][:min(2, npl)], near]).astype(int)
        cand = cand[cand < npl]
        for j in rng.choice(cand, size=min(len(cand), int(rng.integers(1, 4))), replace=False):
            pl = p["planes"][j]
            qc = X[k, 3:] * np.array([-1, -1, -1, 1.0])
            z = np.concatenate([_quat_rot(qc, pl[:3]), [pl[:3] @ X[k, :3] + pl[3]]])
            z[:3] += rng.normal(size=3) * 0.002; z[3] += rng.normal() * nz
            pl_kf.append(k); pl_id.append(int(j)); pl_z.append(z)
    return dict(X0=X0, V0=V0, planes0=planes0, ei=ei, ej=ej, between=np.concatenate([zt, zq], 1), between_info=wi,
                plane_kf=np.array(pl_kf), plane_id=np.array(pl_id), plane_z=np.array(pl_z),
                plane_cov=np.array([1e-4, 0, 0, 1e-4, 0, 1e-4]))


def vio_graph(p, device=0, factors=None):
    """assemble config 4 through the C-ABI.  ids: X(k) = k, V(k) = K + k, B(k) = 2K + k, L(j) = 3K + j"""
    f = vio_factors(p) if factors is None else factors
    X, V, K = p["X"], p["V"], len(p["X"])
    gr = Graph(device=device)
    gr.add_poses(f["X0"])
    for k in range(K):
        lib.fgo_add_vec3(gr._h, K + k, _dp(np.ascontiguousarray(f["V0"][k])))
    zb = np.zeros(6)
    for k in range(K):
        lib.fgo_add_bias(gr._h, 2 * K + k, _dp(zb))
    for j, q in enumerate(f["planes0"]):
        gr.add_plane(3 * K + j, q)
    w = np.zeros(21); w[[0, 6, 11, 15, 18, 20]] = 1e14
    gr.add_prior(0, X[0], w)
    gr.add_prior_vec3(K, V[0], 1e-3)
    gr.add_prior_bias(2 * K, zb, 1e-3)
    gr.set_gravity(p["gravity"])
    gr.add_edges(f["ei"], f["ej"], f["between"], np.tile(f["between_info"], (len(f["ei"]), 1)), tangent_order=FGO_TANGENT_GTSAM)
    import ctypes as C
    ids = np.zeros(6, np.int64)
    for k in range(K - 1):
        ids[:] = [k, K + k, k + 1, K + k + 1, 2 * K + k, 2 * K + k + 1]
        gr._chk(lib.fgo_add_imu_combined(gr._h, ids.ctypes.data_as(C.POINTER(C.c_int64)), _dp(np.ascontiguousarray(p["pre"][k]))))
    for k, j, z in zip(f["plane_kf"], f["plane_id"], f["plane_z"]):
        gr.add_plane_factor(int(k), 3 * K + int(j), z, f["plane_cov"])
    return gr, len(f["plane_kf"])


def _noisy_relative(rng, X, ei, ej, sigma_t, sigma_q):
    qa_c = X[ei, 3:] * np.array([-1, -1, -1, 1.0])
    zt = _quat_rot(qa_c, X[ej, :3] - X[ei, :3]) + rng.normal(size=(len(ei), 3)) * sigma_t
    dq = np.concatenate([rng.normal(size=(len(ei), 3)) * sigma_q, np.ones((len(ei), 1))], 1)
    zq = _quat_mul(_quat_mul(qa_c, X[ej, 3:]), dq / np.linalg.norm(dq, axis=1, keepdims=True))
    zq *= np.sign(zq[:, 3:4]) + (zq[:, 3:4] == 0)
    return np.concatenate([zt, zq], 1)


def _g2o_info(n, sigma_t, sigma_q):
    w = np.zeros(21); w[[0, 6, 11]] = 1.0 / sigma_t ** 2; w[[15, 18, 20]] = 1.0 / sigma_q ** 2
    return np.tile(w, (n, 1))


def torus_graph(nu=320, nv=320, R=40.0, r=12.0, seed=46, sigma_t=0.02, sigma_q=0.005):
    """Non-lattice topology for the ordering / launch heuristics: poses on a torus surface, nu x nv grid,
    every pose tied to its 4 grid neighbours with wrap-around in BOTH directions (genus 1: no planar separator structure like
    the Manhattan walk's) plus one chord across the tube per 16 poses.  g2o semantics; returns the dict layout of
    synth_manhattan3d (poses = noisy start, truth, ei < ej, meas, info)."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, np.arange(nv) * 2 * np.pi / nv, indexing="ij")
    u, v = u.ravel(), v.ravel()
    t = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], 1)
    yaw, pitch = u + np.pi / 2, v
    qz = np.stack([np.zeros_like(yaw), np.zeros_like(yaw), np.sin(yaw / 2), np.cos(yaw / 2)], 1)
    qy = np.stack([np.zeros_like(pitch), np.sin(pitch / 2), np.zeros_like(pitch), np.cos(pitch / 2)], 1)
    X = np.concatenate([t, _quat_mul(qz, qy)], 1)
    idx = np.arange(nu * nv).reshape(nu, nv)
    a = np.concatenate([idx.ravel(), idx.ravel()])
    b = np.concatenate([np.roll(idx, -1, 0).ravel(), np.roll(idx, -1, 1).ravel()])
    ch = np.arange(0, nu * nv, 16)
    cu, cv = ch // nv, ch % nv
    a = np.concatenate([a, ch]); b = np.concatenate([b, idx[cu, (cv + nv // 2) % nv]])
    ei, ej = np.minimum(a, b), np.maximum(a, b)
    keep = ei != ej
    key = np.unique(ei[keep].astype(np.int64) * (nu * nv) + ej[keep])
    ei, ej = (key // (nu * nv)).astype(np.int64), (key % (nu * nv)).astype(np.int64)
    meas = _noisy_relative(rng, X, ei, ej, sigma_t, sigma_q)
    X0 = X.copy(); X0[1:, :3] += rng.normal(size=(len(X) - 1, 3)) * 0.05
    return dict(poses=X0, truth=X, ei=ei, ej=ej, meas=meas, info=_g2o_info(len(ei), sigma_t, sigma_q))


def hub_graph(n=100000, seed=47, sigma_t=0.02, sigma_q=0.005, extra=4):
    """Non-uniform degrees: a random-walk chain in which every pose also ties to `extra` earlier poses drawn by preferential
    attachment within a sliding window of 60 poses plus a dozen global 'places' a thousand poses each revisit (degrees 2 .. 1000+):
    linearisation hubs and arrow-ordered hub columns in g2o semantics at 100k poses."""
    rng = np.random.default_rng(seed)
    step = rng.normal(size=(n, 3)) * 0.3; step[:, 2] *= 0.1
    t = np.cumsum(step, 0)
    yaw = np.cumsum(rng.normal(size=n) * 0.05)
    X = np.concatenate([t, np.stack([np.zeros(n), np.zeros(n), np.sin(yaw / 2), np.cos(yaw / 2)], 1)], 1)
    a = [np.arange(n - 1)]; b = [np.arange(1, n)]
    places = rng.choice(n // 10, 12, replace=False)                      # early poses that keep being revisited
    for k in range(extra):
        j = np.arange(50, n)
        w = np.minimum(j, 60)
        i = j - 1 - (rng.random(len(j)) ** 3 * (w - 1)).astype(np.int64)  # mostly the last few poses, sometimes 60 back
        a.append(i); b.append(j)
    jj = rng.choice(np.arange(n // 10, n), n // 8, replace=False)
    a.append(rng.choice(places, len(jj))); b.append(jj)
    a, b = np.concatenate(a), np.concatenate(b)
    ei, ej = np.minimum(a, b), np.maximum(a, b)
    key = np.unique(ei[ei != ej].astype(np.int64) * n + ej[ei != ej])
    ei, ej = (key // n).astype(np.int64), (key % n).astype(np.int64)
    meas = _noisy_relative(rng, X, ei, ej, sigma_t, sigma_q)
    X0 = X.copy(); X0[1:, :3] += rng.normal(size=(n - 1, 3)) * 0.05
    return dict(poses=X0, truth=X, ei=ei, ej=ej, meas=meas, info=_g2o_info(len(ei), sigma_t, sigma_q))
