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


def compile_cpp(tmp, name):
    build.build()
    exe = os.path.join(tmp, name)
    lib = os.path.join(ROOT, "slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_ccicp_adapter_compiles_against_the_cabi(tmp_path):
    assert os.path.exists(compile_cpp(str(tmp_path), "ccicp_test"))


def _quat_rpy(roll, pitch, yaw):
    cy, sy, cp, sp, cr, sr = (np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2),
                              np.cos(roll / 2), np.sin(roll / 2))
    return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy]


@pytest.mark.gpu
@pytest.mark.parametrize("rtype,tall", [(0, False), (1, False), (0, True)])
def test_ccicp_facade_matches_oracle_chain(tmp_path, rtype, tall):
    """CCICP::doICPMatch(initPose) end to end (icpTools.cpp:222-298): two 131 k-ray clouds in, pose out,
    against the same chain made of oracle pieces.  tall: a few returns 70 m above the sensor make the scene's voxel lattice
    larger than the device-resident chain's accumulator (2 M voxels): the adapter redoes that match through the stepwise
    entry points -- the same numbers must come out."""
    exe = compile_cpp(str(tmp_path), "ccicp_test")
    d = str(tmp_path)
    ka, kb = 3, 4
    A, pa = synth.make_cloud3d(ka, n_loop=50)
    B, pb = synth.make_cloud3d(kb, n_loop=50)
    if tall:
        # clusters (a polar bin keeps its points only from six on) far out and high up (overhead returns): 265 x 245 x 37 voxels of 0.5 x 0.5 x 2 m
        rs = np.random.RandomState(2)
        extra = [c + rs.uniform(-0.1, 0.1, (10, 3)) for c in ([66.0, 61.0, 30.0], [-66.0, -61.0, 28.0], [66.0, -61.0, 25.0], [-66.0, 61.0, 31.0], [3.0, 4.0, 72.0])]
        B = np.ascontiguousarray(np.concatenate([B] + extra).astype(np.float32))
    ca, sa = np.cos(pa[2]), np.sin(pa[2])
    rel = np.array([ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1])])
    rel_th = pb[2] - pa[2]
    init = [rel[0] + 0.15, rel[1] - 0.1, 0.05] + _quat_rpy(0.0, 0.0, rel_th + 0.03)
    lab_a, *_ = O.gseg_segment(A)
    out_a, gnd_a = A[lab_a >= O.GSEG_OBSTACLE], A[lab_a == O.GSEG_GROUND]
    target = out_a if rtype else A
    for name, a in (("target.f32", target), ("scene.f32", B), ("target_ground.f32", gnd_a), ("init.f64", np.array(init))):
        np.ascontiguousarray(a).tofile(os.path.join(d, name))
    out = os.path.join(d, "out.bin")
    subprocess.check_call([exe, d, out, str(rtype)])
    got = np.fromfile(out, np.float64)

    # the same chain from oracle pieces
    fa = O.classify_ga(out_a)
    kept = np.flatnonzero(fa != 255)
    bx = np.floor((out_a[:, 0].astype(np.float64) + 300.0) / 0.5).astype(np.int64)
    by = np.floor((out_a[:, 1].astype(np.float64) + 300.0) / 0.5).astype(np.int64)
    order = kept[np.argsort((bx * 1200 + by)[kept], kind="stable")]
    seg_target = np.concatenate([out_a[order], (fa[order] == 1).astype(np.float32)[:, None]], 1)
    m_ga, m_nga = O.ccicp_split(seg_target, O.ccicp_crop(seg_target, init[0], init[1]))
    lab_b, *_ = O.gseg_segment(B)
    out_b = B[lab_b >= O.GSEG_OBSTACLE]
    fb = O.classify_ga(out_b)
    kb_ = fb != 255
    seg_scene, n_vox = O.voxel_downsample(np.concatenate([out_b[kb_], fb[kb_, None].astype(np.float32)], 1))
    s_ga, s_nga = O.ccicp_split(seg_scene, None)
    gnd_scene, n_gvox = O.voxel_downsample(np.concatenate([B[lab_b == O.GSEG_GROUND], np.zeros((int((lab_b == O.GSEG_GROUND).sum()), 1), np.float32)], 1), (0.5, 0.5, 5.0))
    assert list(got[8:16]) == [len(seg_target), n_vox, len(gnd_a), n_gvox, len(m_ga), len(m_nga), len(s_ga), len(s_nga)]
    assert int(got[16]) == (1 if tall else 0)        # the tall scene went through the stepwise entry points, the others through the chain
    scene_xyz = np.fromfile(out + ".scene", np.float32).reshape(-1, 3)      # getSegmentedClouds: the voxel-filtered scene
    assert len(scene_xyz) == n_vox and np.abs(scene_xyz - seg_scene[:, :3]).max() < 1e-4
    assert len(s_ga) + len(s_nga) > 300 and len(m_ga) + len(m_nga) > 5000

    yaw0 = rel_th + 0.03
    R0, t0 = synth.pose_to_Rt(init[0], init[1], yaw0)
    # the voxel centroids differ in the last float bit between oracle and device: fit the oracle on the oracle's
    model = O.IcpModel(m_ga, m_nga)
    R, t, trace, steps = model.fit(s_ga, s_nga, R0, t0, O.icp_params(20, 1e-6, 5.0))
    yaw = np.arctan2(R[1, 0], R[0, 0])
    assert abs(got[0] - t[0]) < 1e-4 and abs(got[1] - t[1]) < 1e-4
    q = _quat_rpy(0.0, 0.0, yaw)
    assert np.abs(got[3:7] - q).max() < 1e-5
    assert abs(got[7] - trace[-1, 7]) <= 2                                  # correspondences of the last step
    z, nc, _ = O.ccicp_height(gnd_a, [got[0], got[1], init[2]] + list(got[3:7]))
    assert abs(got[2] - z) < 1e-6            # (near the sensor the ring pattern leaves no ground within 3 m: z may stay)
    # and the match is sane: B's pose in A's frame, to the 0.5 m voxel centroids the scene is reduced to
    assert abs(got[0] - rel[0]) < 0.3 and abs(got[1] - rel[1]) < 0.3 and abs(yaw - rel_th) < 0.03


@pytest.mark.gpu
def test_mls_one_cloud_form_matches_oracle(tmp_path):
    """MLS::addToMap(cloud, pose) as local_mapper calls it (local_mapper.cpp:107; mls.cpp:34-150): setPose, the cloud
    turned into the global orientation, segmentGround inside, the drv and ground loops in their order, global_cloud;
    then offsetMap and filterPointCloud -- against the oracle's segmentation and in-order grid update."""
    exe = compile_cpp(str(tmp_path), "mls_cloud_test")
    d = str(tmp_path)
    clouds, poses = [], []
    for k in range(3):
        xyz, p = synth.make_cloud3d(k, n_loop=50)
        clouds.append(xyz)
        yaw = 0.02 * k
        poses.append([0.13 * k, -0.21 * k, 0.0] + _quat_rpy(0.0, 0.0, yaw))
        xyz.tofile(os.path.join(d, "cloud%d.f32" % k))
    np.array(poses).tofile(os.path.join(d, "poses.f64"))
    out = os.path.join(d, "out.bin")
    subprocess.check_call([exe, d, out, "3"])
    raw = open(out, "rb").read()
    head = np.frombuffer(raw[:64], np.float64)
    occ = np.frombuffer(raw[64:64 + 40000], np.int8)
    cloud_out = np.frombuffer(raw[64 + 40000:], np.float32).reshape(-1, 3)

    res, size = 0.2, 200
    gp = O.grid_params(size, size, res, min_cluster_points=20, rolling=1)
    num, drv, eocc = np.zeros(size * size), np.full(size * size, -1, np.int8), np.full(size * size, -1, np.int8)
    cx = cy = 0.0
    gc = np.zeros((0, 3), np.float32)
    n_drv = n_gnd = 0
    for k in range(3):
        px, py = poses[k][0], poses[k][1]
        dx, dy = int(np.round((px - cx) / res)), int(np.round((py - cy) / res))     # mls.cpp:419-424
        if dx or dy:
            from test_gpu_stream import roll
            num = roll(num.reshape(size, size), dx, dy).reshape(-1)
            drv = roll(drv.reshape(size, size), dx, dy, -1).reshape(-1)
            eocc = roll(eocc.reshape(size, size), dx, dy, -1).reshape(-1)
            cx += dx * res
            cy += dy * res
            gc = gc + np.array([-(dx * res), -(dy * res), 0], np.float32)             # :433-454
            crop = np.float32(size * res / 2)
            gc = gc[(gc[:, 0] >= -crop) & (gc[:, 0] <= crop) & (gc[:, 1] >= -crop) & (gc[:, 1] <= crop)]
        yaw = 0.02 * k
        c, s = np.cos(yaw), np.sin(yaw)
        # tf's matrix from the quaternion the program was given (not from the angle): the same doubles
        q = poses[k][3:]
        dd = sum(v * v for v in q); s2 = 2.0 / dd
        xs, ys, zs = q[0] * s2, q[1] * s2, q[2] * s2
        wx, wy, wz, xx, xy, xz, yy, yz, zz = q[3] * xs, q[3] * ys, q[3] * zs, q[0] * xs, q[0] * ys, q[0] * zs, q[1] * ys, q[1] * zs, q[2] * zs
        Rm = np.array([[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]])
        P = clouds[k].astype(np.float64)
        T = np.stack([Rm[0, 0] * P[:, 0] + Rm[0, 1] * P[:, 1] + Rm[0, 2] * P[:, 2] + (cx - px),
                      Rm[1, 0] * P[:, 0] + Rm[1, 1] * P[:, 1] + Rm[1, 2] * P[:, 2] + (cy - py),
                      Rm[2, 0] * P[:, 0] + Rm[2, 1] * P[:, 1] + Rm[2, 2] * P[:, 2] + 0.0], 1).astype(np.float32)
        lab, *_ = O.gseg_segment(T)
        o, g = T[lab == O.GSEG_OBSTACLE], T[lab == O.GSEG_GROUND]
        O.grid_add_scan_inorder(gp, np.concatenate([o, np.zeros((len(o), 1), np.float32)], 1),
                                np.concatenate([g, np.zeros((len(g), 1), np.float32)], 1), num, drv, eocc)
        gc = np.concatenate([gc, o])
        n_drv, n_gnd = len(o), len(g)
    assert (head[0], head[1]) == (cx, cy)
    assert (int(head[2]), int(head[3])) == (n_drv, n_gnd)
    assert np.array_equal(occ, eocc)
    assert int(head[4]) == len(gc)                                      # global_cloud before the filter
    gc = gc + np.array([0, 0, 0.25], np.float32)                        # offsetMap(z = 0.25)
    # filterPointCloud(0.1, 0.1): one centroid per voxel; the set of occupied voxels and their centroids
    assert 0 < len(cloud_out) < len(gc)
    key = lambda a: np.floor(a / np.float32(0.1)).astype(np.int64)
    want = {tuple(k) for k in key(gc)}
    assert len(cloud_out) == len(want)
    assert abs(float(cloud_out[:, 2].mean()) - float(gc[:, 2].mean())) < 0.05
