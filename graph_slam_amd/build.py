"""Build recipes: libfgo.so (HIP, gfx950) and the host-side C++ mirror of the reference interface.

Everything is built IN-TREE with explicit hipcc / g++ command lines (no JIT cache), so the
artefacts travel with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "graph_slam_amd")
CSRC = os.path.join(PKG, "csrc")
LIBFGO = os.path.join(PKG, "libfgo.so")

FGO_SOURCES = ["fgo_core.cpp", "host_alloc.cpp", "fgo_structure.cpp", "fgo_lm.cpp", "fgo_isam2.cpp", "fgo_dist.cpp", "fgo_inspect.cpp", "synth.cpp", "ordering.cpp", "symbolic.cpp", "imu_preint.cpp", "kernels.hip", "kernels_gtsam.hip", "kernels_ba.hip", "preint_kernel.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP toolchain is required to build libfgo.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_libfgo(force=False, verbose=True):
    """One object file per source (compiled in parallel, only when the source or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in FGO_SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "fgo.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-result", "-fvisibility=hidden", "-DFGO_LOCAL_OPERATORS"]
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=ROOT)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    vmap = os.path.join(CSRC, "libfgo.map")                 # the C-ABI (fgo_*) is all the library exports
    if jobs or _stale(LIBFGO, objs + [vmap]):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-Wl,--version-script=" + vmap, "-o", LIBFGO] + objs)
    return LIBFGO


def build_oracle(verbose=True):
    """Builds the CPU restatement (test infrastructure).  Building the checker is not using it."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", odir] + ([] if verbose else ["-s"]), check=True)
    return os.path.join(odir, "liborc.so")


def build_host(verbose=True):
    """C++ mirror of the reference's CGraphG2O surface + example driver (links against libfgo.so)."""
    hdir = os.path.join(PKG, "host")
    mk = os.path.join(hdir, "Makefile")
    if not os.path.exists(mk):
        return None
    subprocess.run(["make", "-C", hdir] + ([] if verbose else ["-s"]), check=True)
    return hdir


if __name__ == "__main__":
    build_libfgo(force="--force" in sys.argv)
    build_oracle()
    build_host()
