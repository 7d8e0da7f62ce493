"""The model index built by the device kernels (slam_amd/csrc/icp_build.hip) is the same bytes as the
single-threaded host build it replaces -- cell index and halo lists -- on every model shape the ICP tests use,
and a handle made from device-resident arrays (slam_icp_create_dev) registers exactly like one made from host
arrays.  Stands where the reference copies the model and builds its kd-trees (icp.cpp:51-69, kdtree.cpp:72-106)."""
import time

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import api, synth

pytestmark = pytest.mark.gpu


def both(m_ga, m_nga, **kw):
    dev = api.Icp(m_ga, m_nga, **kw)
    host = api.Icp(m_ga, m_nga, build_on_host=1, **kw)
    assert dev.build_info()[0] and not host.build_info()[0]
    return dev, host


def assert_same_index(dev, host):
    a, b = dev.index_info(), host.index_info()
    assert a == b, (a, b)
    for which in (0, 1):
        x, y = dev.index_blob(which), host.index_blob(which)
        assert x.shape == y.shape and (which == 1 or x.size > 0)
        diff = np.flatnonzero(x != y)
        assert diff.size == 0, "blob %d differs at %d bytes, first at offset %d of %d" % (which, diff.size, diff[0], x.size)


def models():
    rs = np.random.RandomState(7)
    m_ga, m_nga = synth.make_map()
    yield "config 2 map (pillars GA, walls NGA)", m_ga, m_nga, {}
    a, b = synth.make_map(all_nga=True)
    yield "all NGA", a, b, {}
    yield "index kept in HBM", m_ga, m_nga, {"force_global": 1}
    yield "ring search only (no lists)", m_ga, m_nga, {"lanes_per_point": 8}
    yield "five points", np.zeros((0, 2)), rs.rand(5, 2) * 3, {}
    gx, gy = np.meshgrid(np.arange(40) * 0.25, np.arange(30) * 0.25)
    grid = np.stack([gx.ravel(), gy.ravel()], 1)
    yield "gridded map with duplicates (equal keys, equal cells)", grid[:300], np.concatenate([grid, grid[:100]]), {}
    bad = m_nga.copy()
    bad[5] = [np.nan, 1.0]
    bad[17] = [2.0, np.inf]
    bad[40] = [-np.inf, np.nan]
    bad[41] = [1e30, np.nan]
    yield "non-finite points", m_ga, bad, {}
    yield "all points equal", np.zeros((0, 2)), np.tile([[3.0, -2.0]], (500, 1)), {}
    yield "one wall (collinear)", np.zeros((0, 2)), np.stack([np.linspace(-20, 20, 4000), np.full(4000, 1.5)], 1), {}
    big = rs.randn(19999, 2) * [30.0, 20.0]
    big2 = rs.rand(19999, 2) * [80.0, 60.0] - [40.0, 30.0]
    yield "2 x 19999 points (the CCICP cap, icpTools.h:21)", big, big2, {}
    dense = rs.randn(6000, 2) * 0.4     # hundreds of points per cell near the centre, lists still fit
    yield "dense cluster", dense[:1000], dense[1000:], {}
    yield "coarse forced pitch", m_ga, m_nga, {"cell_size": 1.3}
    blob = rs.randn(6000, 2) * 0.004 + [2.0, 1.0]   # one list cell holds thousands of entries: past the LDS sort
    yield "thousands of points in a centimetre", blob[:500], np.concatenate([blob[500:], rs.rand(50, 2) * 10]), {}
    wall = np.stack([rs.rand(3000) * 0.5, 3.0 + rs.randn(3000) * 0.002], 1)   # a wall seen by a thousand scans
    yield "dense wall segment (long lists, sorted in LDS)", wall[:100], np.concatenate([wall[100:], m_nga[:3000]]), {}


@pytest.mark.parametrize("case", list(models()), ids=lambda c: c[0])
def test_device_build_is_the_host_build(case):
    _, m_ga, m_nga, kw = case
    dev, host = both(m_ga, m_nga, **kw)
    assert_same_index(dev, host)
    dev.close()
    host.close()


def test_create_dev_registers_like_create():
    """slam_icp_create_dev (model resident in HBM): same index bytes, same registration, and both equal the oracle."""
    m_ga, m_nga = synth.make_map()
    d_ga, d_nga = api.DeviceArray.from_host(m_ga, np.float64), api.DeviceArray.from_host(m_nga, np.float64)
    a = api.Icp.from_device(d_ga, len(m_ga), d_nga, len(m_nga), max_iter=12, min_delta=-1.0)
    b = api.Icp(m_ga, m_nga, max_iter=12, min_delta=-1.0, build_on_host=1)
    for which in (0, 1):
        assert np.array_equal(a.index_blob(which), b.index_blob(which))
    batch = synth.make_batch(8, n_loop=256)
    Ra, ta, ra, _ = a.fit_batch(batch)
    Rb, tb, rb, _ = b.fit_batch(batch)
    assert np.array_equal(ra["n_corr"], rb["n_corr"])
    # the centroid the sums are shifted by comes from a tree reduction there, a running sum here: last-bit differences
    assert np.abs(ta - tb).max() < 1e-9 and np.abs(Ra - Rb).max() < 1e-9
    model = O.IcpModel(m_ga, m_nga)
    Ro, to, _, nc, _ = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t, O.icp_params(12, -1.0, 5.0))
    assert np.array_equal(ra["n_corr"], nc) and np.abs(ta - to).max() < 1e-4 and np.abs(Ra - Ro).max() < 1e-5
    a.close()
    b.close()


def test_create_is_fast_and_pooled(capsys):
    """slam_icp_create for 10 k points < 0.3 ms and for 2 x 19 999 < 1 ms (wall clock of the
    whole call, after the first call has filled the library's buffer pool)."""
    rs = np.random.RandomState(3)
    m_ga, m_nga = synth.make_map()
    big = (rs.randn(19999, 2) * [30.0, 20.0], rs.rand(19999, 2) * [80.0, 60.0] - [40.0, 30.0])
    out = {}
    for name, (ga, nga), limit in (("10k", (m_ga, m_nga), 0.3e-3), ("2x19999", big, 1.0e-3)):
        api.Icp(ga, nga).close()
        best, info = 1.0, None
        for _ in range(10):
            t0 = time.perf_counter()
            icp = api.Icp(ga, nga)
            dt = time.perf_counter() - t0
            if dt < best:
                best, info = dt, icp.build_info()[1]
            icp.close()
        out[name] = (best, info)
        assert best < limit * 1.5, (name, best, info)   # the bound with a margin for a shared host; bench.py reports the number
    with capsys.disabled():
        for k, (best, info) in out.items():
            print("\n  slam_icp_create %s: %.3f ms (enqueueing the build %.3f, its one wait %.3f)" % (k, best * 1e3, info[0], info[1]))


def test_builds_from_two_threads_at_once():
    """Two host threads creating handles at the same time (the reference runs two CCICP objects in one process,
    scan_registration.cpp:57 and graphSlamTools.cpp:14): the builds share the library's build stream, its buffer pool and
    its pinned plan blocks; every handle still holds the bytes of the host build of its own model."""
    import threading
    rs = np.random.RandomState(5)
    m_ga, m_nga = synth.make_map()
    models = [(m_ga, m_nga), (rs.randn(3000, 2) * 4.0, rs.rand(9000, 2) * [50.0, 20.0])]
    want = []
    for ga, nga in models:
        host = api.Icp(ga, nga, build_on_host=1)
        want.append((host.index_info(), host.index_blob(0), host.index_blob(1)))
        host.close()
    errors = []

    def work(k):
        try:
            ga, nga = models[k]
            for _ in range(25):
                dev = api.Icp(ga, nga)
                ok = dev.index_info() == want[k][0] and np.array_equal(dev.index_blob(0), want[k][1]) and np.array_equal(dev.index_blob(1), want[k][2])
                dev.close()
                if not ok:
                    errors.append("thread %d: a build differs from the host build of its model" % k)
                    return
        except Exception as e:      # noqa: BLE001 (reported below)
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
