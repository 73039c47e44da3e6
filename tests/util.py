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
