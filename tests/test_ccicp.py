"""The CCICP facade steps either side of the ICP: voxel filter
(icpTools.cpp:620-633), crop + GA/NGA split with the cap (:225-276) and the height recovery (:301-381).
The reference does them with PCL, which is not in its checkout: oracle/ccicp_oracle.c restates the
published PCL 1.7 algorithms (parity unpinned); the HIP path is compared with it through the C-ABI --
bit-exact for counts, order, indices and the split arrays, 1e-5 m for voxel centroids (PCL's own sums are
order-dependent float; oracle sums in double, the device in 64-bit fixed point) and 1e-6 m for z."""
import numpy as np
import pytest

import oracle_lib as O
from slam_amd import synth


def obstacle_cloud(k=3):
    xyz = synth.make_cloud3d(k, n_loop=50)[0]
    lab, *_ = O.gseg_segment(xyz)
    obs = xyz[lab >= O.GSEG_OBSTACLE]
    return xyz, lab, obs


def test_oracle_voxel_grid_properties():
    xyz, lab, obs = obstacle_cloud()
    flags = O.classify_ga(obs)
    keep = flags != 255
    pts = np.concatenate([obs[keep], flags[keep, None].astype(np.float32)], axis=1)
    out, n = O.voxel_downsample(pts)
    assert 0 < n < len(pts) / 3
    # one output per occupied voxel, in increasing voxel index (x fastest, then y, then z)
    inv = np.float32(1) / np.array([0.5, 0.5, 2.0], np.float32)
    ijk = np.floor(out[:, :3] * inv).astype(np.int64)
    key = (ijk[:, 2] - ijk[:, 2].min()) * 10**8 + (ijk[:, 1] - ijk[:, 1].min()) * 10**4 + (ijk[:, 0] - ijk[:, 0].min())
    assert (np.diff(key) > 0).all()
    vin = np.floor(pts[:, :3] * inv).astype(np.int64)
    assert len(np.unique(vin, axis=0)) == n
    # the mean of the centroids weighted by population is the mean of the cloud
    _, inverse, counts = np.unique(vin, axis=0, return_inverse=True, return_counts=True)
    assert counts.sum() == len(pts)
    # 0.5 m classification bins coincide with the voxel columns: the averaged flag is the bin's flag
    assert set(np.unique(out[:, 3])) <= {0.0, 1.0}
    assert O.voxel_downsample(np.zeros((0, 4), np.float32))[1] == 0
    bad = pts[:10].copy()
    bad[3, 0] = np.nan
    assert O.voxel_downsample(bad)[1] == O.voxel_downsample(np.delete(bad, 3, axis=0))[1]
    assert O.voxel_downsample(np.array([[0, 0, 0, 0], [1e9, 1e9, 1e9, 0]], np.float32), (0.01, 0.01, 0.01))[1] == -1


def test_oracle_crop_split_and_cap():
    rs = np.random.RandomState(3)
    pts = np.concatenate([(rs.rand(5000, 3) * [400, 400, 2] - [200, 200, 1]), rs.randint(0, 2, (5000, 1))], 1).astype(np.float32)
    keep = O.ccicp_crop(pts, 10.0, -20.0)
    x_lo, x_hi = np.float32(-75 + 10.0), np.float32(75 + 10.0)
    assert np.array_equal(keep, (pts[:, 0] >= x_lo) & (pts[:, 0] <= x_hi) & (pts[:, 1] >= np.float32(-95)) & (pts[:, 1] <= np.float32(55)))
    edge = np.array([[x_hi, 0, 0, 1], [np.nextafter(x_hi, np.float32(1e9)), 0, 0, 1], [0, 0, np.inf, 0]], np.float32)
    assert list(O.ccicp_crop(edge, 10.0, -20.0)) == [True, False, False]     # closed interval; non-finite points go
    ga, nga = O.ccicp_split(pts, keep)
    sel = pts[keep]
    assert np.array_equal(ga, sel[sel[:, 3] > 0.5][:, :2].astype(np.float64))
    assert np.array_equal(nga, sel[sel[:, 3] <= 0.5][:, :2].astype(np.float64))
    ga, nga = O.ccicp_split(pts, None, cap=101)                             # ICP_MAX_PTS - 1 per class, cloud order
    assert len(ga) == 100 and len(nga) == 100
    assert np.array_equal(ga, pts[pts[:, 3] > 0.5][:100, :2].astype(np.float64))


def test_oracle_height_interpolate():
    xyz, lab, obs = obstacle_cloud(5)
    ground = xyz[lab == O.GSEG_GROUND]
    # the lowest ring meets the ground 3.7 m out, so look under a robot standing 5 m ahead of the sensor
    pose = [5.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    z, nc, idx = O.ccicp_height(ground, pose)
    assert nc == 4 and all(i >= 0 for i in idx)
    # flat ground 1.73 m below the sensor: z = n_z * 1.45 + mean ground z
    assert abs(z - (1.45 + synth.GROUND_Z)) < 0.03
    z2, nc2, _ = O.ccicp_height(ground[:0], pose)
    assert nc2 == 0 and z2 == pose[2]
    far = ground + np.float32([100, 0, 0])
    z3, nc3, _ = O.ccicp_height(far, pose)
    assert nc3 < 4 and z3 == pose[2]                                        # "Height could not be determined" (:379)
    # a tilted plane z = 0.1 x - 1.5 seen from a yawed pose
    gx, gy = np.meshgrid(np.arange(-3, 3, 0.05), np.arange(-3, 3, 0.05))
    plane = np.stack([gx.ravel(), gy.ravel(), 0.1 * gx.ravel() - 1.5], 1).astype(np.float32)
    yaw = 0.7
    z4, nc4, _ = O.ccicp_height(plane, [0.2, -0.1, 0.0, 0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
    n_z = 1 / np.sqrt(1 + 0.01)
    assert nc4 == 4 and abs(z4 - (n_z * 1.45 + (0.1 * 0.2 - 1.5))) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("k,leaf", [(3, (0.5, 0.5, 2.0)), (9, (0.5, 0.5, 5.0)), (12, (0.25, 0.3, 1.0))])
def test_gpu_voxel_downsample_matches_oracle(k, leaf):
    from slam_amd import api
    xyz, lab, obs = obstacle_cloud(k)
    seg = api.GroundSegmentation()
    flags = seg.classify_ga(obs)
    cc = api.Ccicp()
    out = cc.voxel_downsample(obs, flags, leaf)
    keep = flags != 255
    pts = np.concatenate([obs[keep], flags[keep, None].astype(np.float32)], axis=1)
    ref, n = O.voxel_downsample(pts, leaf)
    assert len(out) == n > 100
    assert np.array_equal(out[:, 3], ref[:, 3])
    assert np.abs(out[:, :3] - ref[:, :3]).max() < 1e-5
    # (same voxels in the same order: the rows pair up one to one above)
    # the flag can also come in as the fourth float; non-finite and dropped points do not count
    pts2 = np.concatenate([pts, [[np.nan, 0, 0, 1], [0, np.inf, 0, 0]]]).astype(np.float32)
    out2 = cc.voxel_downsample(pts2, None, leaf)
    assert np.array_equal(out2, out)
    assert len(cc.voxel_downsample(np.zeros((0, 3), np.float32))) == 0
    with pytest.raises(api.SlamError):
        cc.voxel_downsample(np.array([[0, 0, 0], [1e6, 1e6, 1e3]], np.float32), None, (0.01, 0.01, 0.01))
    seg.close(); cc.close()


@pytest.mark.gpu
def test_gpu_split_matches_oracle_and_feeds_icp():
    from slam_amd import api
    cc = api.Ccicp()
    rs = np.random.RandomState(5)
    pts = np.concatenate([(rs.rand(60000, 3) * [400, 400, 2] - [200, 200, 1]), rs.randint(0, 2, (60000, 1))], 1).astype(np.float32)
    pts[17, 1] = np.nan
    for pose_xy, cap in ((None, 20000), ((10.0, -20.0), 20000), ((10.0, -20.0), 301), ((1e4, 0.0), 20000)):
        ga, nga = cc.split(pts, pose_xy, 75.0, cap)
        keep = O.ccicp_crop(pts, *pose_xy) if pose_xy is not None else None
        rga, rnga = O.ccicp_split(pts, keep, cap)
        assert np.array_equal(ga, rga, equal_nan=True) and np.array_equal(nga, rnga, equal_nan=True)
    assert len(ga) == 0 and len(nga) == 0                                   # everything cropped away
    # cloud -> segmentation -> classification -> voxel filter -> split -> ICP model, all through the C-ABI
    xyz, lab, obs = obstacle_cloud(7)
    seg = api.GroundSegmentation()
    flags = seg.classify_ga(obs)
    vox = cc.voxel_downsample(obs, flags)
    m_ga, m_nga = cc.split(vox, (0.0, 0.0))
    assert len(m_ga) + len(m_nga) == len(vox) > 500
    icp = api.Icp(m_ga, m_nga, max_iter=5)
    R, t, res = icp.fit(m_ga[:200], m_nga[:400], np.eye(2), np.array([0.05, -0.03]))
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, trace, steps = model.fit(m_ga[:200], m_nga[:400], np.eye(2), np.array([0.05, -0.03]), O.icp_params(5, 1e-6, 5.0))
    assert np.abs(t - to).max() < 1e-9 and res.iters == steps
    icp.close(); seg.close(); cc.close()


@pytest.mark.gpu
def test_gpu_height_matches_oracle():
    from slam_amd import api
    cc = api.Ccicp()
    xyz, lab, obs = obstacle_cloud(5)
    ground = xyz[lab == O.GSEG_GROUND]
    rs = np.random.RandomState(2)
    for trial in range(6):
        yaw, pitch = rs.uniform(-3, 3), rs.uniform(-0.05, 0.05)
        q = np.array([0, np.sin(pitch / 2), 0, np.cos(pitch / 2)])         # small pitch, then yaw
        qz = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
        quat = [qz[3] * q[0] + qz[0] * q[3] + qz[1] * q[2] - qz[2] * q[1],
                qz[3] * q[1] - qz[0] * q[2] + qz[1] * q[3] + qz[2] * q[0],
                qz[3] * q[2] + qz[0] * q[1] - qz[1] * q[0] + qz[2] * q[3],
                qz[3] * q[3] - qz[0] * q[0] - qz[1] * q[1] - qz[2] * q[2]]
        pose = [5.0 + rs.uniform(-1, 1), rs.uniform(-2, 2), rs.uniform(-0.2, 0.2)] + quat
        z, nc, idx = cc.height(ground, pose)
        zo, nco, idxo = O.ccicp_height(ground, pose)
        assert (nc, idx) == (nco, idxo) and (trial > 0 or nc == 4)
        assert abs(z - zo) < 1e-6
    z, nc, idx = cc.height(ground + np.float32([100, 0, 0]), [0, 0, 0.3, 0, 0, 0, 1])
    assert nc < 4 and z == 0.3
    z, nc, idx = cc.height(np.zeros((0, 3), np.float32), [0, 0, 0.3, 0, 0, 0, 1])
    assert nc == 0 and z == 0.3 and idx == [-1] * 4
    cc.close()


@pytest.mark.gpu
def test_gpu_bin_order_is_classify_points_order():
    """classifyPoints rebuilds the cloud bin by bin (icpTools.cpp:64-101): x bin major, y bin minor, original
    order inside a bin, edge cells and outside points dropped -- the order the ICP_MAX_PTS cap then cuts."""
    from slam_amd import api
    xyz, lab, obs = obstacle_cloud(6)
    rs = np.random.RandomState(8)
    pts = np.concatenate([obs, (rs.rand(3000, 3) * [700, 700, 2] - [350, 350, 1]).astype(np.float32)])
    pts = pts[rs.permutation(len(pts))]
    seg, cc = api.GroundSegmentation(), api.Ccicp()
    flags = seg.classify_ga(pts)
    assert np.array_equal(flags, O.classify_ga(pts))
    out = cc.bin_order(pts, flags)
    bx = np.floor((pts[:, 0].astype(np.float64) + 300.0) / 0.5).astype(np.int64)
    by = np.floor((pts[:, 1].astype(np.float64) + 300.0) / 0.5).astype(np.int64)
    kept = np.flatnonzero(flags != 255)
    order = kept[np.argsort((bx * 1200 + by)[kept], kind="stable")]
    assert len(out) == len(order) and 0 < len(out) < len(pts)
    assert np.array_equal(out[:, :3], pts[order])
    assert np.array_equal(out[:, 3], (flags[order] == 1).astype(np.float32))
    # the cap cuts in that order
    ga, nga = cc.split(out, None, cap=501)
    sel = out[out[:, 3] > 0.5][:500, :2].astype(np.float64)
    assert np.array_equal(ga, sel)
    assert len(cc.bin_order(np.zeros((0, 3), np.float32), np.zeros(0, np.uint8))) == 0
    seg.close(); cc.close()


@pytest.mark.gpu
def test_gpu_select_outcloud_in_cloud_order():
    """labels -> the outcloud CCICP::segmentGround classifies (obstacle + overhead) and its ground cloud
    (icpTools.cpp:106-119), cloud order kept."""
    import ctypes as C
    from slam_amd import api
    xyz = synth.make_cloud3d(4, n_loop=50, rings=32, n_az=1024)[0]
    n = len(xyz)
    seg, cc = api.GroundSegmentation(), api.Ccicp()
    d_xyz = api.DeviceArray.from_host(xyz)
    d_lab = api.DeviceArray((n,), np.uint8)
    d_out = api.DeviceArray((n, 4), np.float32)
    seg.segment_dev(d_xyz, n, 3, d_lab)
    api.synchronize()
    lab = d_lab.download()
    for mask, want in (((1 << api.GSEG_OBSTACLE) | (1 << api.GSEG_OVERHEAD), xyz[lab >= api.GSEG_OBSTACLE]),
                       (1 << api.GSEG_GROUND, xyz[lab == api.GSEG_GROUND]), (0, xyz[:0])):
        cnt = C.c_int(-1)
        api.check(api.lib().slam_ccicp_select_dev(cc.h, d_xyz.ptr, n, 3, d_lab.ptr, mask, d_out.ptr, C.byref(cnt), None))
        got = d_out.download()[:cnt.value]
        assert cnt.value == len(want) and np.array_equal(got[:, :3], want) and (got[:, 3] == 0).all()
    seg.close(); cc.close()
