"""Builds the HIP library in-tree for gfx950: slam_amd/lib/libslam_mi355x.so
(hipcc cross-compiles without a GPU).  `python -m slam_amd.build [--force] [--measure]`.

--measure additionally builds slam_amd/lib/libslam_mi355x_measure.so with -DSLAM_MEASURE: in-kernel stamps,
the raycast ablation switches, the SLAM_ICP_* / SLAM_RAYCAST_* environment knobs and the slam_icp_debug_*
entry points of include/slam_mi355x_measure.h (tools/ scripts load it with SLAM_AMD_MEASURE=1).  The shipped
library has none of them."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libslam_mi355x.so")
MEASURE_LIB = os.path.join(LIBDIR, "libslam_mi355x_measure.so")
RCCL_LIB = os.path.join(LIBDIR, "libslam_mi355x_rccl.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["runtime.hip", "icp.hip", "icp_build.hip", "icp_single.hip", "grid.hip", "gseg.hip", "ccicp.hip",
           "mapper.hip"]
RCCL_SOURCES = ["rccl.hip"]
# -ffp-contract=off: the reference arithmetic (x86-64, no FMA) rounds every
# product before the add; the kernels additionally spell the parity-critical
# expressions with __fmul_rn/__fadd_rn.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _headers():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    d += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")]
    return d


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _build_lib(lib, objdir, extra, force, verbose):
    """One hipcc per stale source, side by side, then one link."""
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = _headers()
    jobs, objs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if jobs or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC"] + objs + ["-o", lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return True
    return False


def build(force=False, verbose=False, measure=False):
    os.makedirs(LIBDIR, exist_ok=True)
    built = []
    if _build_lib(LIB, os.path.join(LIBDIR, "obj"), [], force, verbose):
        built.append(LIB)
    extra = ["-DSLAM_MEASURE"] + os.environ.get("SLAM_MEASURE_DEFINES", "").split()   # e.g. -DSLAM_SEED_RING=1 for an A/B
    if measure and _build_lib(MEASURE_LIB, os.path.join(LIBDIR, "obj_measure"), extra, force, verbose):
        built.append(MEASURE_LIB)
    rccl_srcs = [os.path.join(CSRC, s) for s in RCCL_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if rccl_srcs and (force or _stale(RCCL_LIB, rccl_srcs + _headers() + [LIB])):
        # RCCL is resolved at load time: in a torch process torch's bundled librccl (same
        # SONAME) is already mapped and is the one used; otherwise /opt/rocm/lib's.
        cmd = [HIPCC] + FLAGS + ["-shared"] + rccl_srcs + \
              ["-o", RCCL_LIB, "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lrccl",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L" + LIBDIR, "-l:libslam_mi355x.so"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(RCCL_LIB)
    return built


if __name__ == "__main__":
    out = build(force="--force" in sys.argv, verbose=True, measure="--measure" in sys.argv)
    print("built:" if out else "up to date:", out or [LIB])
