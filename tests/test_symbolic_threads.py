"""Host structure phase (ordering.cpp + symbolic.cpp) on several host threads: the elimination order, the update lists
and the task / panel schedule must be the same for every thread count (the parallel top of the nested dissection emits
in post-order, the update lists are generated per target column with no shared cursors).  CPU only: builds the
developer tool tools/symstats.cpp, which links the two sources without the device part."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "symstats")


@pytest.fixture(scope="module")
def symstats():
    src = [os.path.join(ROOT, "tools", "symstats.cpp")] + [os.path.join(ROOT, "graph_slam_amd", "csrc", f)
                                                          for f in ("symbolic.cpp", "ordering.cpp", "synth.cpp")]
    if not os.path.exists(EXE) or any(os.path.getmtime(f) > os.path.getmtime(EXE) for f in src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + ROOT, "-o", EXE] + src, check=True, cwd=os.path.join(ROOT, "tools"))
    return EXE


def run(exe, n, threads, leaf=64):
    env = dict(os.environ, FGO_HOST_THREADS=str(threads))
    out = subprocess.run([exe, str(n), "5", "4", str(leaf), "5000", "1000000000"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    keep = [l for l in out.stdout.splitlines() if l.startswith(("nnzL", "panels", "ops:", "level 0:", "structure checksum", "critical-path"))]
    assert any(l.startswith("structure checksum") for l in keep)
    return keep


@pytest.mark.parametrize("n", [600, 20000])
def test_structure_does_not_depend_on_the_thread_count(symstats, n):
    ref = run(symstats, n, 1)
    for threads in (2, 5, 16):
        assert run(symstats, n, threads) == ref


def test_small_leaf_size_exercises_the_parallel_top_tree(symstats):
    # leaf 8: the top tree is expanded even on a small graph (regions > 8 * leaf are bisected concurrently)
    assert run(symstats, 3000, 8, leaf=8) == run(symstats, 3000, 1, leaf=8)


def test_ordering_quality_on_the_benchmark_graph(symstats):
    """cfg 2 (100k poses / 1M edges): the level-synchronous sweeps pay ~70 us per level, so the depth of the schedule is
    the quantity the separator selection is tuned for (69 levels with raw level sizes from one end, 35 now).
    Bounds with some slack: a change that costs levels or fill should be a decision, not an accident."""
    import re
    line = [l for l in run(symstats, 100000, 4) if l.startswith("nnzL")][0]
    nnz = int(re.search(r"nnzL blocks (\d+)", line).group(1))
    nops = int(re.search(r"nops (\d+)", line).group(1))
    height = int(re.search(r"etree_height (\d+)", line).group(1))
    levels = int(re.search(r"levels (\d+)", line).group(1))
    assert levels <= 40 and height <= 460
    assert nnz <= 2.45e6 and nops <= 34.0e6


def test_fgo_tune_overrides_a_schedule_constant(symstats):
    """The tuning constants are overridden through ONE variable, FGO_TUNE="key=value,..." (csrc/fgo_internal.hpp tune()):
    merge_multi=0 restores round 1's panel rule (a panel does not continue across a separator boundary), which costs levels;
    an unknown key is ignored."""
    import re

    def levels(tune):
        env = dict(os.environ, FGO_HOST_THREADS="4")
        if tune is not None:
            env["FGO_TUNE"] = tune
        out = subprocess.run([symstats, "20000", "5", "4", "64", "5000", "1000000000"], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return int(re.search(r"levels (\d+)", [l for l in out.stdout.splitlines() if l.startswith("nnzL")][0]).group(1))
    base = levels(None)
    assert levels("no_such_key=3") == base
    assert levels("merge_multi=0,no_such_key=1") > base
