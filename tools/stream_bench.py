"""BASELINE config 5 timing: a stream of 1081-beam scans through the C++ StreamMapper
(pinned host chunks -> H2D -> ICP -> rolling 2000^2 @ 0.05 m grid), pipelined on three
HIP streams vs one stage after another.  This is the PCIe-inclusive rate quoted in DESIGN.md.
    python tools/stream_bench.py [n_scans] [chunk_scans] [repeat]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import build, synth  # noqa: E402


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    build.build()
    d = tempfile.mkdtemp()
    lib = os.path.join(ROOT, "slam_amd", "lib")
    exe = os.path.join(d, "stream_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "stream_test.cpp"), "-o", exe,
                           "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    m_ga, m_nga = synth.make_map(10000)
    base = synth.make_batch(256, n_loop=256)
    reps = (n_scans + 255) // 256
    pts = np.concatenate([base.pts] * reps)
    off = np.concatenate([[0]] + [base.scan_off[1:] + r * base.n_points for r in range(reps)]).astype(np.int32)
    nga = np.concatenate([base.scan_nga] * reps)
    R = np.concatenate([base.R] * reps)
    t = np.concatenate([base.t] * reps)
    off = off[:n_scans + 1]
    pts = pts[:off[-1]]
    for name, a in (("m_ga.f64", m_ga), ("m_nga.f64", m_nga), ("pts.f64", pts), ("scan_off.i32", off),
                    ("scan_nga.i32", nga[:n_scans]), ("R0.f64", R[:n_scans]), ("t0.f64", t[:n_scans])):
        np.ascontiguousarray(a).tofile(os.path.join(d, name))
    for c in ([chunk] if len(sys.argv) > 2 else [16, 64, 256]):
        sys.stdout.write(subprocess.check_output([exe, d, os.path.join(d, "out.bin"), str(c), "2000", "0.05",
                                                  str(repeat)]).decode())
        sys.stdout.flush()


if __name__ == "__main__":
    main()
