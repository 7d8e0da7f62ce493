"""The two ROS node shims (ros/scan_registration_node.cpp, ros/local_mapper_node.cpp: scan_registration.cpp:57-199,
local_mapper.cpp:29-130 over the library) compile against stub message headers with the real messages' type and field
names and link against the C-ABI library.  Compile-only: there is no ROS in this image."""
import os
import subprocess

import pytest

from slam_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("node", ["scan_registration_node", "local_mapper_node"])
def test_ros_shim_compiles_and_links(tmp_path, node):
    build.build()
    lib = os.path.join(ROOT, "slam_amd", "lib")
    exe = str(tmp_path / node)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                           "-I", os.path.join(ROOT, "tests", "cpp", "ros_stub"), "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "ros"), os.path.join(ROOT, "ros", node + ".cpp"), "-o", exe,
                           "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)
