"""Builds the HIP library in-tree for gfx950: slam_amd/lib/libslam_mi355x.so
(hipcc cross-compiles without a GPU).  `python -m slam_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libslam_mi355x.so")
RCCL_LIB = os.path.join(LIBDIR, "libslam_mi355x_rccl.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["runtime.hip", "icp.hip", "grid.hip", "gseg.hip", "ccicp.hip"]
RCCL_SOURCES = ["rccl.hip"]
# -ffp-contract=off: the reference arithmetic (x86-64, no FMA) rounds every
# product before the add; the kernels additionally spell the parity-critical
# expressions with __fmul_rn/__fadd_rn.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(srcs):
    d = [os.path.join(CSRC, s) for s in srcs]
    d += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    d += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))
          if f.endswith(".h")]
    return d


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    built = []
    if force or _stale(LIB, _deps(SOURCES)):
        # one hipcc per source, side by side (icp.hip alone is most of the time), then one link
        objdir = os.path.join(LIBDIR, "obj")
        os.makedirs(objdir, exist_ok=True)
        cflags = [f for f in FLAGS if f != "-shared"]
        jobs = []
        for s in SOURCES:
            obj = os.path.join(objdir, s.replace(".hip", ".o"))
            cmd = [HIPCC] + cflags + ["-c", os.path.join(CSRC, s), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, obj, subprocess.Popen(cmd)))
        for cmd, obj, proc in jobs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
        cmd = [HIPCC, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC"] + [j[1] for j in jobs] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(LIB)
    rccl_srcs = [s for s in RCCL_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if rccl_srcs and (force or _stale(RCCL_LIB, _deps(rccl_srcs))):
        # RCCL is resolved at load time: in a torch process torch's bundled librccl (same
        # SONAME) is already mapped and is the one used; otherwise /opt/rocm/lib's.
        cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in rccl_srcs] + \
              ["-o", RCCL_LIB, "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lrccl",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L" + LIBDIR, "-l:libslam_mi355x.so"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(RCCL_LIB)
    return built


if __name__ == "__main__":
    out = build(force="--force" in sys.argv, verbose=True)
    print("built:" if out else "up to date:", out or [LIB])
