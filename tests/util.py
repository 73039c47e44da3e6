"""Shared numpy helpers for the tests (independent of both the product and the oracle)."""
import numpy as np


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + bw * ax + ay * bz - az * by,
                     aw * by + bw * ay + az * bx - ax * bz,
                     aw * bz + bw * az + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_rot(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ v


def pose_mul(a, b):
    q = quat_mul(a[3:], b[3:])
    return np.concatenate([a[:3] + quat_rot(a[3:], b[:3]), q / np.linalg.norm(q)])


def pose_inv(a):
    qc = a[3:] * np.array([-1, -1, -1, 1])
    return np.concatenate([-quat_rot(qc, a[:3]), qc])


def random_pose(rng, scale=1.0):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(size=3) * scale, q])


def info_full(ut):
    W = np.zeros((6, 6)); k = 0
    for r in range(6):
        for c in range(r, 6):
            W[r, c] = W[c, r] = ut[k]; k += 1
    return W


def info_ut(W):
    return np.array([W[r, c] for r in range(6) for c in range(r, 6)])


def random_info(rng, lo=50.0, hi=400.0):
    """random SPD 6x6 with a dense off-diagonal structure"""
    A = rng.normal(size=(6, 6))
    Q, _ = np.linalg.qr(A)
    return Q @ np.diag(rng.uniform(lo, hi, size=6)) @ Q.T


def noisy(rng, pose, st, sq):
    d = np.concatenate([rng.normal(size=3) * st, rng.normal(size=3) * sq])
    w = np.sqrt(max(0.0, 1 - d[3:] @ d[3:]))
    return pose_mul(pose, np.concatenate([d[:3], d[3:], [w]]))


def small_graph(rng, n=10, extra=10, noise=0.03, dense_info=True, fixed_first=True):
    """random-walk pose graph: chain + `extra` random loop closures; returns kwargs for orc.Problem"""
    truth = [np.array([0, 0, 0, 0, 0, 0, 1.0])]
    for _ in range(1, n):
        step = np.concatenate([rng.normal(size=3) * 0.5 + [1, 0, 0], [0, 0, 0, 1]])
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.7, 0.7)
        step[3:] = np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]])
        truth.append(pose_mul(truth[-1], step))
    truth = np.array(truth)
    ei, ej = [], []
    for k in range(1, n):
        ei.append(k - 1); ej.append(k)
    seen = set(zip(ei, ej))
    tries = 0
    while len(ei) < n - 1 + extra and tries < 100 * (extra + 1):
        tries += 1
        a, b = sorted(rng.integers(0, n, size=2))
        if a == b or (a, b) in seen:
            continue
        seen.add((a, b)); ei.append(int(a)); ej.append(int(b))
    meas, info = [], []
    for a, b in zip(ei, ej):
        z = pose_mul(pose_inv(truth[a]), truth[b])
        meas.append(noisy(rng, z, noise, noise * 0.5))
        W = random_info(rng) if dense_info else np.diag([1 / noise ** 2] * 3 + [4 / noise ** 2] * 3)
        info.append(info_ut(W))
    poses = [truth[0].copy()]
    for k in range(1, n):
        poses.append(pose_mul(poses[-1], meas[k - 1]))      # odometry chaining (g2o_graph.cpp:118)
    fixed = np.zeros(n, np.uint8)
    if fixed_first:
        fixed[0] = 1
    return dict(poses=np.array(poses), fixed=fixed, ei=np.array(ei, np.int32), ej=np.array(ej, np.int32),
                meas=np.array(meas), info=np.array(info))


# ---- mixed-variable GTSAM-semantics scenario (poses + plane landmarks + 3D points) ---------------------------------
SR4000_CALIB = np.array([250.5773, 250.5773, 0.0, 90.0, 70.0, -0.8466, 0.5370, 0.0, 0.0])   # test_ba_imu_graph.cpp:84


def mixed_graph(rng, n_poses=8, n_planes=3, n_points=12, obs_per_point=3, noise=0.01):
    """Returns a dict describing a small VIO/BA-like graph in GTSAM semantics:
    values (N x 7), vkind (N), ei/ej/kind/meas(7)/info(21) per binary factor, priors (ids, mean7, info21), calib, bps.
    Variable order: poses, planes, points.  Uses the oracle's factor functions to synthesise noise-free measurements."""
    from tests import orc_binding as orc
    g = small_graph(rng, n=n_poses, extra=n_poses // 2, noise=noise, fixed_first=False)
    truth = [np.array([0, 0, 0, 0, 0, 0, 1.0])]
    for k in range(n_poses - 1):                       # rebuild a smooth ground truth by chaining the measurements
        truth.append(pose_mul(truth[-1], g["meas"][k]))
    truth = np.array(truth)
    N = n_poses + n_planes + n_points
    values = np.zeros((N, 7)); values[:, 6] = 1.0
    vkind = np.zeros(N, np.int32)
    values[:n_poses] = [noisy(rng, t, 0.05, 0.02) for t in truth]
    values[0] = truth[0]
    ei, ej, kind, meas, info = [], [], [], [], []
    Wb = np.diag([1 / 0.01 ** 2] * 3 + [1 / 0.02 ** 2] * 3)
    for a, b in zip(g["ei"], g["ej"]):
        z = pose_mul(pose_inv(truth[a]), truth[b])
        ei.append(int(a)); ej.append(int(b)); kind.append(orc.FK_BETWEEN)
        meas.append(noisy(rng, z, noise, noise * 0.5)); info.append(info_ut(Wb))
    # plane landmarks (gtsam_graph.cpp:1198-1206: Sigma = diag(1e-4))
    for p in range(n_planes):
        vid = n_poses + p
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        pl = np.array([n[0], n[1], n[2], rng.uniform(2.0, 6.0)])
        values[vid, :4] = orc.plane_retract(pl, rng.normal(size=3) * 0.05); values[vid, 4:] = 0
        vkind[vid] = orc.VK_PLANE
        for k in rng.choice(n_poses, size=min(n_poses, 4), replace=False):
            z = orc.plane_retract(orc.plane_transform(pl, truth[k]), rng.normal(size=3) * noise)
            ei.append(int(k)); ej.append(vid); kind.append(orc.FK_PLANE)
            m = np.zeros(7); m[:4] = z; meas.append(m)
            w = np.zeros(21); w[:6] = [1e4, 0, 0, 1e4, 0, 1e4]; info.append(w)
    # points seen through the distorted camera (gtsam_graph.cpp:373-409)
    bps = np.concatenate([[0.05, -0.02, 0.1], [0.5, 0.5, 0.5, 0.5]])        # body_P_sensor: a 120-degree axis swap + offset
    prior_ids, prior_mean, prior_info = [0], [truth[0]], [info_ut(np.diag([1e14] * 6))]
    for q in range(n_points):
        vid = n_poses + n_planes + q
        seen = rng.choice(n_poses, size=min(n_poses, obs_per_point), replace=False)
        cam = pose_mul(truth[seen[0]], bps)
        pc = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), rng.uniform(2.0, 5.0)])
        pw = cam[:3] + quat_rot(cam[3:], pc)
        vkind[vid] = orc.VK_POINT
        values[vid, :3] = pw + rng.normal(size=3) * 0.02; values[vid, 3:] = 0
        for k in seen:
            ck = pose_mul(truth[k], bps)
            pk = quat_rot(ck[3:] * np.array([-1, -1, -1, 1]), pw - ck[:3])                # point in camera k
            if pk[2] < 0.5 or abs(pk[0] / pk[2]) > 0.45 or abs(pk[1] / pk[2]) > 0.45:     # inside the SR4000's field of view
                continue                                                                  # (the k1/k2 model folds over beyond it)
            r = orc.reproj(truth[k], pw, np.zeros(2), SR4000_CALIB, bps, jac=False)      # projection (uv = 0)
            ei.append(int(k)); ej.append(vid); kind.append(orc.FK_REPROJ)
            m = np.zeros(7); m[:2] = r + rng.normal(size=2) * 0.5; meas.append(m)
            w = np.zeros(21); w[0] = 1.0; info.append(w)                                  # Isotropic::Sigma(2, 1.0)
        prior_ids.append(vid)                                                              # PriorFactor<Point3> sigma 0.014
        pm = np.zeros(7); pm[:3] = values[vid, :3]; prior_mean.append(pm)
        pw6 = np.zeros((6, 6)); pw6[:3, :3] = np.eye(3) / 0.014 ** 2; prior_info.append(info_ut(pw6))
    return dict(values=values, vkind=vkind, ei=np.array(ei, np.int32), ej=np.array(ej, np.int32),
                kind=np.array(kind, np.int32), meas=np.array(meas), info=np.array(info),
                prior_ids=np.array(prior_ids, np.int32), prior_mean=np.array(prior_mean), prior_info=np.array(prior_info),
                calib=SR4000_CALIB.copy(), bps=bps, n_poses=n_poses, n_planes=n_planes, n_points=n_points)


def mixed_oracle(g):
    from tests import orc_binding as orc
    p = orc.Problem(g["values"], np.zeros(len(g["values"]), np.uint8), g["ei"], g["ej"], g["meas"], g["info"])
    p.set_kinds(g["vkind"], g["kind"])
    p.set_calibration(g["calib"], g["bps"])
    p.add_priors(g["prior_ids"], g["prior_mean"], g["prior_info"])
    if g.get("imu_ids") is not None and len(g["imu_ids"]):
        p.add_imu_factors(g["imu_ids"], g["imu_pre"], g["imu_info"])
    return p


def vio_graph(rng, n_kf=8, samples=40, noise=0.01, with_planes=True):
    """VIO-like scenario in GTSAM semantics (test_vro_imu_graph.cpp:191-198, gtsam_graph.cpp:320-368, 613-695):
    keyframe poses X, velocities V, biases B; CombinedImuFactor between consecutive keyframes (40 samples at 200 Hz),
    BetweenFactor<Pose3> from VO, priors on X0 (sigma 1e-7), V0 and B0 (sigma 1e-3), optional plane landmarks.
    Variable order: X_0..X_{K-1}, V_0.., B_0.., planes.  The truth is generated by chaining the preintegrated
    prediction, so every IMU residual vanishes at the truth."""
    from tests import orc_binding as orc
    K = n_kf
    n_planes = 2 if with_planes else 0
    N = 3 * K + n_planes
    values = np.zeros((N, 7)); vkind = np.zeros(N, np.int32)
    vkind[K:2 * K] = orc.VK_VEC3; vkind[2 * K:3 * K] = orc.VK_BIAS; vkind[3 * K:] = orc.VK_PLANE
    bias_true = np.array([0.03, -0.02, 0.01, 0.002, -0.001, 0.0015])
    X = [np.array([0, 0, 0, 0, 0, 0, 1.0])]; V = [np.array([0.3, 0.1, 0.0])]
    pre = []
    for k in range(K - 1):
        t = np.arange(samples) * 0.005
        gyro = np.stack([0.4 * np.sin(2.1 * t + p) for p in rng.uniform(0, 6, 3)], 1) + bias_true[3:]
        acc = np.stack([1.0 * np.cos(1.3 * t + p) for p in rng.uniform(0, 6, 3)], 1) + np.array([0, 0, -9.71]) + bias_true[:3]
        pim = orc.Preint(np.zeros(6), acc, gyro, 0.005)          # integrated with bias estimate 0, true bias != 0
        xj, vj = pim.predict(X[-1], V[-1], bias_true)
        X.append(xj); V.append(vj); pre.append(pim)
    X = np.array(X); V = np.array(V)
    values[:K] = [noisy(rng, x, 0.05, 0.02) for x in X]; values[0] = X[0]
    values[K:2 * K, :3] = V + rng.normal(size=(K, 3)) * 0.05; values[K:2 * K, 6] = 0
    values[2 * K:3 * K, :6] = 0.0; values[2 * K:3 * K, 6] = 0                  # biases start at zero (gtsam_graph.cpp:354)
    values[K, :3] = V[0]
    ei, ej, kind, meas, info = [], [], [], [], []
    Wb = np.diag([1 / 0.01 ** 2] * 3 + [1 / 0.02 ** 2] * 3)
    for k in range(K - 1):
        z = pose_mul(pose_inv(X[k]), X[k + 1])
        ei.append(k); ej.append(k + 1); kind.append(orc.FK_BETWEEN); meas.append(noisy(rng, z, noise, noise * 0.5)); info.append(info_ut(Wb))
    for p in range(n_planes):
        vid = 3 * K + p
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        pl = np.array([n[0], n[1], n[2], rng.uniform(2.0, 6.0)])
        values[vid, :4] = orc.plane_retract(pl, rng.normal(size=3) * 0.05); values[vid, 4:] = 0
        for k in range(0, K, 2):
            z = orc.plane_retract(orc.plane_transform(pl, X[k]), rng.normal(size=3) * noise)
            ei.append(k); ej.append(vid); kind.append(orc.FK_PLANE)
            m = np.zeros(7); m[:4] = z; meas.append(m)
            w = np.zeros(21); w[:6] = [1e4, 0, 0, 1e4, 0, 1e4]; info.append(w)
    w3 = np.zeros((6, 6)); w3[:3, :3] = np.eye(3) / 1e-3 ** 2
    prior_ids = [0, K, 2 * K]
    pm_v = np.zeros(7); pm_v[:3] = V[0]
    prior_mean = [X[0], pm_v, np.zeros(7)]
    prior_info = [info_ut(np.diag([1e14] * 6)), info_ut(w3), info_ut(np.eye(6) / 1e-3 ** 2)]
    imu_ids = [[k, K + k, k + 1, K + k + 1, 2 * K + k, 2 * K + k + 1] for k in range(K - 1)]
    imu_info = [np.linalg.inv(p.cov) for p in pre]
    return dict(values=values, vkind=vkind, ei=np.array(ei, np.int32), ej=np.array(ej, np.int32), kind=np.array(kind, np.int32),
                meas=np.array(meas), info=np.array(info), prior_ids=np.array(prior_ids, np.int32), prior_mean=np.array(prior_mean),
                prior_info=np.array(prior_info), calib=SR4000_CALIB.copy(), bps=np.array([0, 0, 0, 0, 0, 0, 1.0]),
                imu_ids=np.array(imu_ids, np.int32), imu_pre=pre, imu_info=np.array(imu_info), n_kf=K, n_planes=n_planes,
                truth_X=X, truth_V=V, bias_true=bias_true)
