"""The reference-shaped C++ adapters (include/slam_amd/icp.hpp, mls.hpp): a C++
program written like ccicp2d/src/icpTools.cpp:168-197 and
local_mapper.cpp:29,86,107 is compiled with g++, run on the GPU box, and its
results are compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_adapter_test(tmp):
    build.build()
    exe = os.path.join(tmp, "adapter_test")
    lib = os.path.join(ROOT, "slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp"), "-o", exe,
                           "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adapters_compile_against_the_cabi(tmp_path):
    """Not a GPU test: the headers are valid C++ against the shipped library."""
    assert os.path.exists(compile_adapter_test(str(tmp_path)))


@pytest.mark.gpu
def test_adapters_match_oracle(tmp_path):
    exe = compile_adapter_test(str(tmp_path))
    d = str(tmp_path)
    m_ga, m_nga = synth.make_map(6000)
    batch = synth.make_batch(1, n_loop=256)
    t_ga, t_nga = batch.scan(0)
    x, y, th = synth.init_pose(0, batch.true_poses[0])
    rs = np.random.RandomState(4)
    obs = (rs.randn(5000, 4) * [3, 3, 1, 1] + [2, 1, 0, 0]).astype(np.float32)
    gnd = (rs.randn(8000, 4) * 8).astype(np.float32)
    for name, a in (("m_ga.f64", m_ga), ("m_nga.f64", m_nga), ("t_ga.f64", t_ga), ("t_nga.f64", t_nga),
                    ("init.f64", np.array([x, y, th])), ("obs.f32", obs), ("gnd.f32", gnd)):
        np.ascontiguousarray(a).tofile(os.path.join(d, name))
    out = os.path.join(d, "out.bin")
    subprocess.check_call([exe, d, out])
    raw = open(out, "rb").read()
    res = np.frombuffer(raw[:32], np.float64)
    flag = np.frombuffer(raw[32:40], np.float64)[0]
    meta = np.frombuffer(raw[40:72], np.float64)
    occ = np.frombuffer(raw[72:], np.int8)

    model = O.IcpModel(m_ga, m_nga)
    R0, t0 = synth.pose_to_Rt(x, y, th)
    R, t, trace, steps = model.fit(t_ga, t_nga, R0, t0, O.icp_params(20, 1e-6, 5.0))
    assert abs(res[0] - t[0]) < 1e-4 and abs(res[1] - t[1]) < 1e-4
    assert abs(res[2] - np.arctan2(R[1, 0], R[0, 0])) < 1e-5
    assert int(res[3]) == int(trace[-1, 7])
    assert flag == 1.0                                  # icp.cpp:38-43 behaviour
    assert tuple(meta) == (0.2, 200.0, -20.0, -20.0)    # mls.h:167-175
    gp = O.grid_params(200, 200, 0.2, min_cluster_points=20, rolling=1)
    num = np.zeros(40000)
    drv = np.full(40000, -1, np.int8)
    eocc = np.full(40000, -1, np.int8)
    for _ in range(3):
        O.grid_add_scan_inorder(gp, obs, gnd, num, drv, eocc)
    assert np.array_equal(occ, eocc)
