"""Ground-segmentation pre-filter (BASELINE config 3):
groundSegmentation.cpp:110-468 restated in oracle/gseg_oracle.c (CPU tests) and
the HIP path through the C-ABI against it (GPU tests).  Parity is unpinned by the
reference (no tests there, PCL/Eigen absent here); the GP values go through a
Cholesky solve instead of Eigen's inverse, so labels are compared exactly and
the per-bin GP values to 1e-9."""
import numpy as np
import pytest

import oracle_lib as O
from slam_amd import synth


@pytest.fixture(scope="module")
def cloud():
    return synth.make_cloud3d(3, n_loop=50)[0]


def test_oracle_labels_are_plausible(cloud):
    lab, bins, state, value, iters = O.gseg_segment(cloud)
    g = np.abs(cloud[:, 2] - synth.GROUND_Z) < 0.05
    assert (lab[g] == O.GSEG_GROUND).mean() > 0.99          # the floor is found
    assert (lab[g] >= O.GSEG_OBSTACLE).sum() == 0
    high = cloud[:, 2] > synth.GROUND_Z + 0.5
    assert (lab[high] == O.GSEG_GROUND).sum() == 0          # nothing half a metre up is ground
    assert (lab[high] == O.GSEG_OBSTACLE).sum() > 1000 and (lab == O.GSEG_OVERHEAD).sum() > 1000
    assert iters >= 72 // 2
    # model bins carry the prototype height (:385-398), which is the lowest z of the bin
    for b in np.flatnonzero(state == 1)[:200]:
        assert value[b] == cloud[bins == b, 2].min()


def test_oracle_binning_rules():
    # :126 3-D range gate, :129-131 angle wrap, :141 0.5 m range bins, :206 "more than 5 points"
    p = O.gseg_params()
    pts = np.array([[10, 0, -1.7], [0, 10, -1.7], [-10, 0, -1.7], [0, -10, -1.7], [99, 0, 20], [0.2, -0.0001, -1.7]],
                   np.float32)
    lab, bins, state, value, it = O.gseg_segment(pts, p)
    assert list(bins[:4]) == [0 * 200 + 20, 18 * 200 + 20, 36 * 200 + 20, 54 * 200 + 20]
    assert bins[4] == -1                      # sqrt(99^2 + 20^2) > 100
    assert bins[5] == 71 * 200 + 0            # just below the +x axis wraps to the last sector
    assert (lab == O.GSEG_DROPPED).all()      # no bin has more than 5 points
    six = np.tile(np.array([[10, 0.1, -1.7]], np.float32), (6, 1)) + np.arange(6)[:, None] * np.float32(1e-3)
    lab, *_ = O.gseg_segment(six, p)
    assert (lab == O.GSEG_GROUND).all()       # one signal bin, one seed: "model too small" but still labelled (:385)


def test_oracle_ties_and_empty_input():
    lab, bins, state, value, it = O.gseg_segment(np.zeros((0, 3), np.float32))
    assert len(lab) == 0 and it == 0
    flat = synth.make_cloud3d(0, n_loop=50, rings=16, n_az=512)[0].copy()
    flat[np.abs(flat[:, 2] - synth.GROUND_Z) < 0.05, 2] = synth.GROUND_Z     # exact ties in height
    lab, *_ = O.gseg_segment(flat)
    g = flat[:, 2] == np.float32(synth.GROUND_Z)
    assert (lab[g] == O.GSEG_GROUND).mean() > 0.95


@pytest.mark.gpu
@pytest.mark.parametrize("k,rings,n_az", [(3, 64, 2048), (11, 64, 2048), (0, 32, 1024), (7, 16, 512)])
def test_gpu_labels_match_oracle(k, rings, n_az):
    from slam_amd import api
    xyz = synth.make_cloud3d(k, n_loop=50, rings=rings, n_az=n_az)[0]
    seg = api.GroundSegmentation()
    lab = seg.segment(xyz)
    ref, bins, state, value, iters = O.gseg_segment(xyz)
    st, val, it = seg.read_model()
    assert np.array_equal(st, state)
    assert it.sum() == iters
    m = st > 0
    assert np.abs(val[m] - value[m]).max() < 1e-9
    assert np.array_equal(lab, ref)
    seg.close()


@pytest.mark.gpu
def test_gpu_strides_params_and_edge_inputs():
    from slam_amd import api
    xyz = synth.make_cloud3d(5, n_loop=50, rings=32, n_az=1024)[0]
    padded = np.zeros((len(xyz), 8), np.float32)       # PCL PointXYZGD: 32-byte records
    padded[:, :3] = xyz
    padded[:, 3:] = 7.0
    seg = api.GroundSegmentation()
    assert np.array_equal(seg.segment(padded), seg.segment(xyz))
    assert len(seg.segment(np.zeros((0, 3), np.float32))) == 0
    weird = xyz.copy()
    weird[:5, 2] = np.nan
    weird[5:10, 0] = np.inf
    ref, *_ = O.gseg_segment(weird)
    assert np.array_equal(seg.segment(weird), ref)
    seg.close()
    p = dict(gp_groundthreshold=0.15, robotheight=0.8, num_seedpoints=4, seeding_maxrange=20.0)
    seg2 = api.GroundSegmentation(**p)
    ref2, *_ = O.gseg_segment(xyz, O.gseg_params(p_tg=0.15, robot_height=0.8, num_seedpoints=4, max_seed_range=20.0))
    assert np.array_equal(seg2.segment(xyz), ref2)
    seg2.close()


@pytest.mark.gpu
def test_config3_pipeline_segment_split_update():
    """config 3 shape, three clouds: ground segmentation -> drv / ground clouds -> the
    local_mapper update (mls.cpp:59-150), all resident on the device."""
    from slam_amd import api
    seg = api.GroundSegmentation()
    g = api.Grid(200, 200, 0.2, rolling=1, min_cluster_points=20)     # local_mapper.cpp:29,86
    gp = O.grid_params(200, 200, 0.2, min_cluster_points=20, rolling=1)
    eh = np.zeros(40000, np.int32)
    em = np.zeros(40000, np.int32)
    for k in range(3):
        xyz = synth.make_cloud3d(k, n_loop=50)[0]
        n = len(xyz)
        d_xyz = api.DeviceArray.from_host(xyz)
        d_lab = api.DeviceArray((n,), np.uint8)
        d_gnd = api.DeviceArray((n, 4), np.float32)
        d_obs = api.DeviceArray((n, 4), np.float32)
        d_cnt = api.DeviceArray((2,), np.int32)
        seg.segment_dev(d_xyz, n, 3, d_lab)
        seg.split_dev(d_xyz, n, 3, d_lab, d_gnd, d_obs, d_cnt)
        api.synchronize()
        n_gnd, n_obs = d_cnt.download()
        lab = d_lab.download()
        assert (n_gnd, n_obs) == ((lab == api.GSEG_GROUND).sum(), (lab == api.GSEG_OBSTACLE).sum())
        api.check(api.lib().slam_grid_add_endpoints_dev(g.h, d_obs.ptr, int(n_obs), d_gnd.ptr, int(n_gnd), 4, None))
        ref, *_ = O.gseg_segment(xyz)
        O.grid_add_endpoints(gp, xyz[ref == O.GSEG_OBSTACLE], xyz[ref == O.GSEG_GROUND], eh, em)
    api.synchronize()
    hits, misses = g.read_counts()
    assert np.array_equal(hits, eh) and np.array_equal(misses, em)
    assert hits.sum() > 10000 and misses.sum() > 100000
    seg.close(); g.close()


def test_oracle_ga_classification_rules():
    # icpTools.cpp:36-103: 0.5 m cells, GA iff >= 2 of the 8 neighbour cells are empty
    wall = np.stack([np.arange(0, 20, 0.1), np.full(200, 5.2), np.zeros(200)], 1).astype(np.float32)
    f = O.classify_ga(wall)
    assert (f == 1).all()                               # a thin wall: 6 empty neighbours
    gx, gy = np.meshgrid(np.arange(0, 5, 0.25), np.arange(0, 5, 0.25))
    blob = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1).astype(np.float32) + np.float32(0.1)
    f = O.classify_ga(blob)
    inner = (blob[:, 0] > 0.6) & (blob[:, 0] < 4.4) & (blob[:, 1] > 0.6) & (blob[:, 1] < 4.4)
    assert (f[inner] == 0).all() and (f[~inner] == 1).any()
    far = np.array([[299.9, 0, 0], [300.1, 0, 0], [-300.0, 0, 0], [-299.4, 0, 0], [np.nan, 0, 0]], np.float32)
    assert list(O.classify_ga(far)) == [255, 255, 255, 1, 255]   # edge ring and outside are dropped (:60, :72-77)


@pytest.mark.gpu
def test_gpu_ga_classification_matches_oracle():
    from slam_amd import api
    xyz = synth.make_cloud3d(9, n_loop=50)[0]
    seg = api.GroundSegmentation()
    lab = seg.segment(xyz)
    obs = xyz[lab >= api.GSEG_OBSTACLE]                  # obsCloud (icpTools.cpp:114-117)
    rs = np.random.RandomState(1)
    extra = (rs.rand(2000, 3) * [700, 700, 1] - [350, 350, 0]).astype(np.float32)
    pts = np.concatenate([obs, extra])
    assert np.array_equal(seg.classify_ga(pts), O.classify_ga(pts))
    f = seg.classify_ga(obs)
    assert (f == 1).sum() > 100                          # thin walls: every obstacle cell has empty neighbours
    seg.close()
