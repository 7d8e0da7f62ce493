"""The reference-shaped C++ adapters (include/slam_amd/icp.hpp, mls.hpp): a C++
program written like ccicp2d/src/icpTools.cpp:168-197 and
local_mapper.cpp:29,86,107 is compiled with g++, run on the GPU box, and its
results are compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from ccicp_chain import oracle_scan_match, quat_rpy as _quat_rpy
from mls_chain import oracle_local_map
from slam_amd import build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_adapter_test(tmp):
    build.build()
    exe = os.path.join(tmp, "adapter_test")
    lib = os.path.join(ROOT, "slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"),
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
    # slam_amd::IcpPointToPlane (icpPointToPlane.h:26-49): libicp's one-cloud shape and this fork's two-array shape, against the oracle
    res_l = np.fromfile(out + ".p2l", np.float64)
    model_l = O.IcpModel(m_ga, m_nga, normals_k=10)
    Rl, tl, trace_l, steps_l = model_l.fit(t_ga, t_nga, R0, t0, O.icp_params(20, 1e-6, 5.0, O.NN_BRUTE, O.MODE_P2L))
    assert abs(res_l[0] - tl[0]) < 1e-4 and abs(res_l[1] - tl[1]) < 1e-4 and abs(res_l[2] - np.arctan2(Rl[1, 0], Rl[0, 0])) < 1e-5
    assert int(res_l[3]) == len(t_ga) + len(t_nga) and res_l[4] == 1.0
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
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_ccicp_adapter_compiles_against_the_cabi(tmp_path):
    assert os.path.exists(compile_cpp(str(tmp_path), "ccicp_test"))


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

    # the same chain from oracle pieces (tests/ccicp_chain.py)
    e = oracle_scan_match(out_a, gnd_a, B, init)
    assert list(got[8:16]) == e["sizes"]
    assert int(got[16]) == (1 if tall else 0)        # the tall scene went through the stepwise entry points, the others through the chain
    scene_xyz = np.fromfile(out + ".scene", np.float32).reshape(-1, 3)      # getSegmentedClouds: the voxel-filtered scene
    assert len(scene_xyz) == e["sizes"][1] and np.abs(scene_xyz - e["seg_scene"][:, :3]).max() < 1e-4
    assert e["sizes"][6] + e["sizes"][7] > 300 and e["sizes"][4] + e["sizes"][5] > 5000
    yaw = e["yaw"]
    assert abs(got[0] - e["t"][0]) < 1e-4 and abs(got[1] - e["t"][1]) < 1e-4
    assert np.abs(got[3:7] - e["q"]).max() < 1e-5
    assert abs(got[7] - e["n_corr"]) <= 2                                  # correspondences of the last step
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
    snaps = oracle_local_map(clouds, poses, size, res)                  # tests/mls_chain.py: mls.cpp:34-150 from oracle pieces
    cx, cy, eocc, n_drv, n_gnd, gc = snaps[-1]
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


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["seq", "ahead", "batch"])
def test_ccicp_sequence_with_a_target_replacement_matches_the_oracle_chain(tmp_path, form):
    """(form: "seq" = one cloud at a time, the reference's own usage; round 5's throughput forms: "ahead" = the next cloud's scene
    chain on a second stream while the current one is matched, CCICP::prepareSceneCloud; "batch" = CCICP::matchSequence, the
    clouds between two target replacements as one registration batch.  Same oracle, same tolerances.)
    BASELINE config 3 as scan_registration runs it, in small: ten 64-ring clouds through slam_amd::CCICP
    (tests/cpp/ccicp_sequence.cpp: setSceneCloud + doICPMatch per cloud, scan_registration.cpp:139-159), the target
    replaced by the cloud just matched after five (setTargetCloud, :73-104) -- every pose against the oracle chain
    (tests/ccicp_chain.py) for ITS target, before and after the replacement, and the truth errors of both side by side:
    what is left against the truth (centimetres to decimetres) is the chain's own -- 0.5 m voxel centroids, a few hundred
    correspondences -- not the GPU's."""
    exe = compile_cpp(str(tmp_path), "ccicp_sequence")
    d = str(tmp_path)
    n, advance = 10, 5
    clouds, poses = zip(*[synth.make_cloud3d(k, n_loop=50) for k in range(n)])
    init, truth, target_of = [], [], []
    for k in range(1, n):
        j = ((k - 1) // advance) * advance
        pa, pb = poses[j], poses[k]
        ca, sa = np.cos(pa[2]), np.sin(pa[2])
        rel = (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])
        init.append([rel[0] + 0.1, rel[1] - 0.1, 0.0] + _quat_rpy(0.0, 0.0, rel[2] + 0.02))
        truth.append(list(rel))
        target_of.append(j)
    for k, c in enumerate(clouds):
        np.ascontiguousarray(c, np.float32).tofile(os.path.join(d, "cloud%d.f32" % k))
    np.array(init, np.float64).tofile(os.path.join(d, "init.f64"))
    np.array(truth, np.float64).tofile(os.path.join(d, "truth.f64"))
    p = subprocess.run([exe, d, str(n), str(advance), "1", form], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["form"] == form
    assert line["matches"] == n - 1 and line["target_updates"] == 2 and line["target_index_builds"] == 2
    got = np.fromfile(os.path.join(d, "poses_out.f64"), np.float64).reshape(n - 1, 7)
    segmented = {}
    err_gpu, err_oracle = [], []
    for m in range(n - 1):
        j = target_of[m]
        if j not in segmented:
            lab, *_ = O.gseg_segment(clouds[j])
            segmented[j] = (clouds[j][lab >= O.GSEG_OBSTACLE], clouds[j][lab == O.GSEG_GROUND])
        out_j, gnd_j = segmented[j]
        e = oracle_scan_match(out_j, gnd_j, clouds[m + 1], init[m])
        assert abs(got[m, 0] - e["t"][0]) < 1e-4 and abs(got[m, 1] - e["t"][1]) < 1e-4, (m, j)
        assert np.abs(got[m, 3:7] - e["q"]).max() < 1e-5, (m, j)
        z, _, _ = O.ccicp_height(gnd_j, [got[m, 0], got[m, 1], init[m][2]] + list(got[m, 3:7]))
        assert abs(got[m, 2] - z) < 1e-6, (m, j)
        err_gpu.append(np.hypot(got[m, 0] - truth[m][0], got[m, 1] - truth[m][1]))
        err_oracle.append(np.hypot(e["t"][0] - truth[m][0], e["t"][1] - truth[m][1]))
    assert set(target_of) == {0, 5}
    assert np.abs(np.array(err_gpu) - np.array(err_oracle)).max() < 2e-4          # the same error against the truth, match by match
    assert abs(line["mean_xy_error_m"] - float(np.mean(err_oracle))) < 2e-4 and max(err_oracle) < 0.5


@pytest.mark.gpu
def test_ccicp_forms_agree_once_the_scene_chains_replay_as_graphs(tmp_path):
    """matchSequence replays a scene slot's chain as a hipGraph from the slot's third use on: whatever a launch of the chain takes
    from the HOST at enqueue time (a counter, an epoch) is frozen in the replay.  Four passes over twelve clouds -- the last one
    all replays -- against the sequential form and the form with two chains in flight: the same poses (round 6: the epochs of
    the one-launch compactions and of the GA lattice were host counters; 2.6 cm in one of config 3's 49 matches)."""
    exe = compile_cpp(str(tmp_path), "ccicp_sequence")
    d = str(tmp_path)
    n, advance = 12, 6
    clouds, poses = zip(*[synth.make_cloud3d(k, n_loop=50) for k in range(n)])
    init, truth = [], []
    for k in range(1, n):
        j = ((k - 1) // advance) * advance
        pa, pb = poses[j], poses[k]
        ca, sa = np.cos(pa[2]), np.sin(pa[2])
        rel = (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])
        init.append([rel[0] + 0.1, rel[1] - 0.1, 0.0] + _quat_rpy(0.0, 0.0, rel[2] + 0.02))
        truth.append(list(rel))
    for k, c in enumerate(clouds):
        np.ascontiguousarray(c, np.float32).tofile(os.path.join(d, "cloud%d.f32" % k))
    np.array(init, np.float64).tofile(os.path.join(d, "init.f64"))
    np.array(truth, np.float64).tofile(os.path.join(d, "truth.f64"))
    got = {}
    for form in ("seq", "ahead", "batch", "batch+graphs"):
        env = dict(os.environ, SEQ_GRAPHS="1") if form.endswith("graphs") else dict(os.environ)   # (replays are opt-in since round 6)
        p = subprocess.run([exe, d, str(n), str(advance), "4", form.split("+")[0]], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        got[form] = np.fromfile(os.path.join(d, "poses_out.f64"), np.float64).reshape(n - 1, 7)
    assert np.array_equal(got["seq"], got["ahead"])
    for form in ("batch", "batch+graphs"):
        assert np.abs(got["seq"] - got[form]).max() < 1e-9, (form, np.abs(got["seq"] - got[form]).max(axis=1))


@pytest.mark.gpu
def test_ccicp_throughput_forms_on_the_edges(tmp_path):
    """tests/cpp/ccicp_forms_test.cpp: CCICP::prepareSceneCloud with the right and the wrong cloud prepared (poses bit-identical to the
    sequential form, read-outs behind an adopted scene), CCICP::matchSequence over 23 scenes (three batches: 16 at most, and a pose
    whose crop window leaves nothing of the target ends one), a 4-point scene inside a batch (orientation.w == 9999, icpTools.cpp:
    179-184).  The program checks itself against its own sequential run; the sequential form against the oracle is the test above."""
    exe = compile_cpp(str(tmp_path), "ccicp_forms_test")
    d = str(tmp_path)
    n = 24
    clouds, poses = zip(*[synth.make_cloud3d(k, n_loop=50) for k in range(n)])
    init = []
    for k in range(1, n):
        pa, pb = poses[0], poses[k]
        ca, sa = np.cos(pa[2]), np.sin(pa[2])
        rel = (ca * (pb[0] - pa[0]) + sa * (pb[1] - pa[1]), -sa * (pb[0] - pa[0]) + ca * (pb[1] - pa[1]), pb[2] - pa[2])
        init.append([rel[0] + 0.1, rel[1] - 0.1, 0.0] + _quat_rpy(0.0, 0.0, rel[2] + 0.02))
    for k, c in enumerate(clouds):
        np.ascontiguousarray(c, np.float32).tofile(os.path.join(d, "cloud%d.f32" % k))
    np.array(init, np.float64).tofile(os.path.join(d, "init.f64"))
    p = subprocess.run([exe, d, str(n)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1500:]
    assert p.stdout.strip().endswith("OK") and "BAD" not in p.stdout
