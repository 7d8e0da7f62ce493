"""The device-resident chain (slam_ccicp_scene_dev, slam_ccicp_height_pose_dev) is the stepwise entry points without
their host round trips: ground segmentation -> select -> classify -> voxel filter / bin order -> crop + split + cap,
every count left on the device.  Same clouds, same bytes; and a registration run straight from its outputs (the host
never learns the cloud's size) equals one run from host arrays.  The stepwise entry points are what tests/test_ccicp.py
and tests/test_gseg.py hold against the oracle."""
import ctypes as C

import numpy as np
import pytest

from slam_amd import api, synth

pytestmark = pytest.mark.gpu
CAP = 20000


def stepwise(seg, cc, xyz, voxel, crop_xy):
    L = api.lib()
    n = len(xyz)
    d_xyz = api.DeviceArray.from_host(xyz)
    d_lab = api.DeviceArray((n,), np.uint8)
    d_obs, d_gnd, d_out = (api.DeviceArray((n, 4), np.float32) for _ in range(3))
    d_flag = api.DeviceArray((n,), np.uint8)
    seg.segment_dev(d_xyz, n, 3, d_lab)
    n_obs, n_gnd, n_out = C.c_int(0), C.c_int(0), C.c_int(0)
    api.check(L.slam_ccicp_select_dev(cc.h, d_xyz.ptr, n, 3, d_lab.ptr, (1 << 2) | (1 << 3), d_obs.ptr, C.byref(n_obs), None))
    api.check(L.slam_ccicp_select_dev(cc.h, d_xyz.ptr, n, 3, d_lab.ptr, 1 << 1, d_gnd.ptr, C.byref(n_gnd), None))
    api.check(L.slam_gseg_classify_ga_dev(seg.h, d_obs.ptr, n_obs.value, 4, d_flag.ptr, None))
    if voxel:
        api.check(L.slam_ccicp_voxel_downsample_dev(cc.h, d_obs.ptr, d_flag.ptr, n_obs.value, 4, 0.5, 0.5, 2.0, d_out.ptr, n,
                                                    C.byref(n_out), None))
    else:
        api.check(L.slam_ccicp_bin_order_dev(cc.h, d_obs.ptr, d_flag.ptr, n_obs.value, 4, d_out.ptr, C.byref(n_out), None))
    d_ga, d_nga = api.DeviceArray((CAP, 2), np.float64), api.DeviceArray((CAP, 2), np.float64)
    counts = (C.c_int * 2)()
    crop = crop_xy is not None
    api.check(L.slam_ccicp_split_dev(cc.h, d_out.ptr, n_out.value, 4, 1 if crop else 0, crop_xy[0] if crop else 0.0,
                                     crop_xy[1] if crop else 0.0, 75.0, CAP, d_ga.ptr, d_nga.ptr, counts, None))
    return dict(ga=d_ga.download()[:counts[0]], nga=d_nga.download()[:counts[1]], ground=d_gnd.download()[:n_gnd.value],
                n_obs=n_obs.value, n_gnd=n_gnd.value, n_flt=n_out.value)


def chain(seg, cc, xyz, voxel, crop_xy, n_cap=None):
    L = api.lib()
    n = len(xyz)
    n_cap = n_cap or n
    d_xyz = api.DeviceArray((n_cap, 3), np.float32)
    api.check(L.slam_memcpy_h2d(d_xyz.ptr, xyz.ctypes.data, xyz.nbytes, None))
    d_pts = api.DeviceArray((2 * CAP, 2), np.float64)
    d_scan = api.DeviceArray((3,), np.int32)
    d_counts = api.DeviceArray((4,), np.int32)
    d_gnd = api.DeviceArray((n_cap, 4), np.float32)
    crop = crop_xy is not None
    api.check(L.slam_ccicp_scene_dev(cc.h, seg.h, d_xyz.ptr, n, 3, 1 if voxel else 0, 1 if crop else 0,
                                     crop_xy[0] if crop else 0.0, crop_xy[1] if crop else 0.0, 75.0, CAP, d_pts.ptr, d_scan.ptr,
                                     d_gnd.ptr, d_counts.ptr, None))
    api.synchronize()
    scan, counts = d_scan.download(), d_counts.download()
    pts = d_pts.download()
    return dict(ga=pts[:scan[2]], nga=pts[scan[2]:scan[1]], ground=d_gnd.download()[:counts[1]], n_obs=int(counts[0]),
                n_gnd=int(counts[1]), n_flt=int(counts[2]), err=int(counts[3]), dev=(d_pts, d_scan, d_gnd, d_counts))


def clouds():
    yield "64-ring cloud, scene (voxel filter)", synth.make_cloud3d(3, n_loop=50)[0], True, None
    yield "64-ring cloud, target (bin order, crop)", synth.make_cloud3d(0, n_loop=50)[0], False, (0.0, 0.0)
    yield "32-ring cloud, scene, cropped off centre", synth.make_cloud3d(7, n_loop=50, rings=32, n_az=1024)[0], True, (30.0, -20.0)
    rs = np.random.RandomState(5)
    junk = (rs.randn(3000, 3) * [20, 20, 1.0]).astype(np.float32)
    junk[::97] = np.nan
    yield "random points with NaNs", junk, True, None
    yield "five points", (rs.randn(5, 3) * 3).astype(np.float32), True, None
    wide = (rs.rand(4000, 3) * [900.0, 900.0, 3.0] - [450.0, 450.0, 1.0]).astype(np.float32)   # outside the 600 m classify lattice too
    yield "points over 900 m", wide, False, None


@pytest.mark.parametrize("case", list(clouds()), ids=lambda c: c[0])
def test_chain_is_the_stepwise_path(case):
    _, xyz, voxel, crop = case
    seg, cc = api.GroundSegmentation(), api.Ccicp()
    a = stepwise(seg, cc, xyz, voxel, crop)
    b = chain(seg, cc, xyz, voxel, crop)
    assert b["err"] == 0
    assert (a["n_obs"], a["n_gnd"], a["n_flt"]) == (b["n_obs"], b["n_gnd"], b["n_flt"])
    for k in ("ga", "nga", "ground"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), k
    seg.close()
    cc.close()


def test_match_and_height_from_the_chain_without_the_host_knowing_sizes():
    """target = cloud 0 (stepwise, host arrays -> slam_icp_create); scene = cloud 1 through the chain, registered by
    slam_icp_fit_batch_dev from the chain's device arrays, height from the device pose -- against slam_icp_fit on host
    arrays and slam_ccicp_height_dev with the host pose."""
    L = api.lib()
    seg, cc = api.GroundSegmentation(), api.Ccicp()
    (c0, p0), (c1, p1) = synth.make_cloud3d(0, n_loop=50), synth.make_cloud3d(1, n_loop=50)
    tgt = stepwise(seg, cc, c0, False, (0.0, 0.0))
    icp = api.Icp(tgt["ga"], tgt["nga"])                      # max_iter 20, min_delta 1e-6 (icp.cpp:27)
    ca, sa = np.cos(p0[2]), np.sin(p0[2])
    rel = (ca * (p1[0] - p0[0]) + sa * (p1[1] - p0[1]), -sa * (p1[0] - p0[0]) + ca * (p1[1] - p0[1]), p1[2] - p0[2])
    R0, t0 = synth.pose_to_Rt(rel[0] + 0.1, rel[1] - 0.1, rel[2] + 0.02)
    ref = stepwise(seg, cc, c1, True, None)
    Rh, th, resh = icp.fit(ref["ga"], ref["nga"], R0, t0)
    sc = chain(seg, cc, c1, True, None)
    d_pts, d_scan, d_gnd, d_counts = sc["dev"]
    d_R, d_t = api.DeviceArray.from_host(R0.reshape(1, 4), np.float64), api.DeviceArray.from_host(t0.reshape(1, 2), np.float64)
    d_res = api.DeviceArray((1,), api.RESULT_DTYPE)
    icp.fit_batch_dev(d_pts, d_scan, d_scan.view(2, (1,)), 1, d_R, d_t, 5.0, d_res, None, None)
    d_z = api.DeviceArray((2,), np.float64)
    # the height against the TARGET's ground cloud, as scan_registration does (ground of cloud 0, kept on the device)
    g0 = api.DeviceArray.from_host(tgt["ground"])
    n0 = api.DeviceArray.from_host(np.array([len(tgt["ground"])], np.int32))
    api.check(L.slam_ccicp_height_pose_dev(cc.h, g0.ptr, n0.ptr, len(tgt["ground"]), 4, d_R.ptr, d_t.ptr, 0.0, d_z.ptr, None))
    api.synchronize()
    R, t, res = d_R.download().reshape(2, 2), d_t.download().reshape(2), d_res.download()[0]
    assert res["iters"] == resh.iters and res["n_corr"] == resh.n_corr
    assert np.abs(R - Rh).max() < 1e-12 and np.abs(t - th).max() < 1e-12
    assert np.hypot(t[0] - rel[0], t[1] - rel[1]) < 0.25      # scan-to-scan on voxel centroids: the method's accuracy, not the chain's
    yaw = np.arctan2(R[1, 0], R[0, 0])
    z_ref, nc_ref, _ = cc.height(tgt["ground"], (t[0], t[1], 0.0, 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)))
    z, nc = d_z.download()
    assert nc == nc_ref == 4 and z == z_ref
    icp.close()
    seg.close()
    cc.close()


def test_chain_on_one_pair_of_handles_over_clouds_of_many_sizes():
    """The one-launch compactions keep their look-back words and their epoch on the device, the GA lattice its call epoch: nothing is
    cleared between calls.  One pair of handles over clouds whose sizes sit on and around the compaction's block size (1024 items),
    shrink and grow again -- a word a larger cloud left behind is never taken for a smaller one's -- each against the stepwise path."""
    seg, cc = api.GroundSegmentation(), api.Ccicp()
    base = synth.make_cloud3d(2, n_loop=50)[0]
    rs = np.random.RandomState(11)
    for n in (70000, 1024, 1025, 5, len(base), 1023, 2049, 4096, 1, 33000):
        xyz = np.ascontiguousarray(base[rs.permutation(len(base))[:n]])
        voxel = n % 2 == 0
        a = stepwise(seg, cc, xyz, voxel, None if n % 3 else (5.0, -5.0))
        b = chain(seg, cc, xyz, voxel, None if n % 3 else (5.0, -5.0))
        assert b["err"] == 0, n
        assert (a["n_obs"], a["n_gnd"], a["n_flt"]) == (b["n_obs"], b["n_gnd"], b["n_flt"]), n
        for k in ("ga", "nga", "ground"):
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), (n, k)
    seg.close()
    cc.close()
