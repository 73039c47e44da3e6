"""CPU-only checks of the product library: it loads, exports every symbol include/fgo.h declares, the
host-only entry points (synthetic generator, shard helper) behave, and device entry points FAIL LOUDLY
without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import graph_slam_amd as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "fgo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fgo_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(G.lib, s), "libfgo.so does not export %s" % s


def test_version_and_error_string():
    assert b"gfx950" in G.lib.fgo_version()


def test_no_gpu_fails_loudly():
    if G.lib.fgo_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(G.FgoError) as ei:
        G.Graph()
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_synth_cfg1_shape_and_determinism():
    g = G.synth_manhattan3d(1000, 4, 0, seed=42)
    assert len(g["ei"]) == 4984                 # ~5k edges: BASELINE config 1
    assert (g["ei"] < g["ej"]).all()            # edges point old -> new (g2o_graph.cpp:174,196-205)
    g2 = G.synth_manhattan3d(1000, 4, 0, seed=42)
    for k in g:
        np.testing.assert_array_equal(g[k], g2[k])
    g3 = G.synth_manhattan3d(1000, 4, 0, seed=43)
    assert not np.array_equal(g["meas"], g3["meas"])
    # unit quaternions, lattice truth, odometry-chained initial guess
    np.testing.assert_allclose(np.linalg.norm(g["poses"][:, 3:], axis=1), 1, atol=1e-12)
    np.testing.assert_allclose(g["truth"][:, :3], np.round(g["truth"][:, :3]), atol=0)
    steps = np.linalg.norm(np.diff(g["truth"][:, :3], axis=0), axis=1)
    np.testing.assert_allclose(steps, 1.0)
    # information = diag(1/sigma_t^2 x3, 1/sigma_q^2 x3)
    W = g["info"][0]
    assert W[0] == pytest.approx(1 / 0.02 ** 2) and W[20] == pytest.approx(1 / 0.005 ** 2) and W[1] == 0


def test_synth_cfg2_edge_budget():
    g = G.synth_manhattan3d(20000, 5, 4, seed=42)
    e = len(g["ei"])
    assert abs(e - 10 * 20000) < 0.01 * 10 * 20000          # 10 edges / pose -> 1M at 100k poses
    span = g["ej"] - g["ei"]
    assert (span >= 1).all() and span.max() > 100           # real long-range loop closures exist


def test_synth_measurements_consistent_with_truth():
    from tests.util import pose_mul, pose_inv
    g = G.synth_manhattan3d(300, 4, 2, seed=5, sigma_t=0.0, sigma_q=0.0)
    for k in range(0, len(g["ei"]), 17):
        z = pose_mul(pose_inv(g["truth"][g["ei"][k]]), g["truth"][g["ej"][k]])
        s = np.sign(z[3:] @ g["meas"][k][3:])
        np.testing.assert_allclose(g["meas"][k][:3], z[:3], atol=1e-12)
        np.testing.assert_allclose(g["meas"][k][3:] * s, z[3:], atol=1e-12)
    np.testing.assert_allclose(g["poses"][:, :3], g["truth"][:, :3], atol=1e-9)


def test_shard_range_partitions():
    lo, hi = C.c_int64(), C.c_int64()
    for n, w in [(10, 3), (1000000, 8), (5, 8), (0, 2)]:
        cover = []
        for r in range(w):
            assert G.lib.fgo_shard_range(n, r, w, C.byref(lo), C.byref(hi)) == 0
            cover.append((lo.value, hi.value))
        assert cover[0][0] == 0 and cover[-1][1] == n
        for a, b in zip(cover, cover[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in cover]
        assert max(sizes) - min(sizes) <= 1
    assert G.lib.fgo_shard_range(10, 3, 3, C.byref(lo), C.byref(hi)) == -1


def test_preint_batch_refuses_without_gpu_and_validates():
    """the batched preintegration is a GPU entry point: no device -> FGO_ENODEV (no CPU fallback), bad input -> FGO_EINVAL"""
    import ctypes as C
    assert hasattr(G.lib, "fgo_preint_batch")
    sp = np.array([0, 2], np.int64)
    acc = np.zeros((2, 3)); gyro = np.zeros((2, 3))
    params = np.zeros(G.IMU_PARAM_DOUBLES); G.lib.fgo_imu_params_vn100(G._dp(params))
    out = np.zeros((1, G.PREINT_DOUBLES))
    bad = np.array([3, 2], np.int64)
    assert G.lib.fgo_preint_batch(0, 1, G._i64p(bad), G._dp(acc), G._dp(gyro), 0.005, None, G._dp(params), G._dp(out)) == -1
    assert G.lib.fgo_preint_batch(0, 1, G._i64p(sp), G._dp(acc), G._dp(gyro), -1.0, None, G._dp(params), G._dp(out)) == -1
    if G.lib.fgo_device_count() <= 0:
        assert G.lib.fgo_preint_batch(0, 1, G._i64p(sp), G._dp(acc), G._dp(gyro), 0.005, None, G._dp(params), G._dp(out)) == -2


def test_exports_are_the_c_abi_only_and_the_allocator_stays_inside():
    """libfgo.so exports the fgo_* entry points and nothing else of its own -- in particular not its operator new / delete
    (csrc/host_alloc.cpp: the library's allocations forward to the process's global operators and add a huge-page hint; a host
    program must never be routed through them).  FGO_THP=0 switches the hint off: the host-only decomposition runs either way."""
    import subprocess
    import sys
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "graph_slam_amd", "libfgo.so")], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("nm not available")
    names = [l.split()[-1] for l in out.stdout.splitlines() if l.strip()]
    own = [n for n in names if not n.startswith("fgo_") and n not in ("FGO_1", "_init", "_fini", "__bss_start", "_edata", "_end")]
    assert own == [], own
    assert not any(n.startswith("_Zn") or n.startswith("_Zd") for n in names)
    code = ("import numpy as np, graph_slam_amd as G\n"
            "g = G.synth_manhattan3d(20000, 5, 4, seed=3)\n"
            "grp = G.debug_partition(20000, g['ei'], g['ej'], 2)\n"
            "print(int(np.bincount(grp, minlength=3)[2]))\n")
    tops = []
    for thp in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, FGO_THP=thp), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        tops.append(int(r.stdout.split()[-1]))
    assert tops[0] == tops[1] and tops[0] > 0
