#!/usr/bin/env python3
"""MLS::addToMap per cloud through the C++ drop-in (include/slam_amd/mls.hpp) as local_mapper's callback runs it
(local_mapper.cpp:65-130; mls.cpp:34-150): tests/cpp/mls_cloud_test.cpp compiled against the shipped library, ten 64-ring clouds of
config 3, getDrivability every five clouds.  `python tools/mls_time.py [passes]`"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measure(passes=6):
    d = tempfile.mkdtemp(prefix="mls_time_")
    exe = os.path.join(d, "mls_cloud_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mls_cloud_test.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "slam_amd", "lib"), "-l:libslam_mi355x.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "slam_amd", "lib")])
    n = 10
    poses = []
    for k in range(n):
        xyz, p = synth.make_cloud3d(k, n_loop=50)
        xyz.tofile(os.path.join(d, "cloud%d.f32" % k))
        c, s = np.cos(0.01 * k), np.sin(0.01 * k)
        poses.append([0.13 * k, -0.21 * k, 0.0, 0.0, 0.0, np.sin(0.005 * k), np.cos(0.005 * k)])
    np.array(poses).tofile(os.path.join(d, "poses.f64"))
    out = {}
    for name, nocloud in (("with global_cloud", "0"), ("disable_pointcloud", "1")):
        p = subprocess.run([exe, d, os.path.join(d, "out.bin"), str(n), str(passes), "5", nocloud], capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-800:])
        out[name] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    out["what"] = ("slam_amd::MLS::addToMap per cloud (include/slam_amd/mls.hpp: mls.cpp:34-150 -- setPose, the cloud turned into the map's "
                   "frame, segmentGround, the drv and ground loops in the reference's order; global_cloud kept or mls.h:223's "
                   "disable_pointcloud) as local_mapper's callback calls it (local_mapper.cpp:107), ten 131 072-ray clouds from pageable host "
                   "memory, getDrivability every five (:120); wall clock of tests/cpp/mls_cloud_test.cpp against the shipped library")
    return out


def main():
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 6)))


if __name__ == "__main__":
    main()
