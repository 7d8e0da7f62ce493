"""BASELINE config 5: a stream of scans through include/slam_amd/stream_mapper.hpp
(H2D copy, registration and the rolling-window grid update overlapped on three
HIP streams).  The C++ driver checks pipelined == one-stage-after-another
bitwise; here its output is checked against the oracle: poses against
oicp_fit (icp.cpp:80-114), the rolling window against MLS::setPose
(mls.cpp:408-479) + the Bresenham oracle on the same poses."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_stream_test(tmp):
    build.build()
    exe = os.path.join(tmp, "stream_test")
    lib = os.path.join(ROOT, "slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "stream_test.cpp"), "-o", exe,
                           "-L" + lib, "-l:libslam_mi355x.so", "-Wl,-rpath," + lib,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_stream_mapper_compiles_against_the_cabi(tmp_path):
    assert os.path.exists(compile_stream_test(str(tmp_path)))


def roll(plane, dx, dy, fill=0):
    """Window after Grid::shiftOrigin(dx,dy) + the clears of mls.cpp:433-477: cell (i,j) shows old (i+dx, j+dy)."""
    sy, sx = plane.shape
    out = np.full_like(plane, fill)
    xs = np.arange(sx) + dx
    ys = np.arange(sy) + dy
    vx = (xs >= 0) & (xs < sx)
    vy = (ys >= 0) & (ys < sy)
    out[np.ix_(vy, vx)] = plane[np.ix_(ys[vy], xs[vx])]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_scans,chunk,size,res", [(48, 8, 600, 0.1), (21, 5, 400, 0.15)])
def test_stream_matches_oracle(tmp_path, n_scans, chunk, size, res):
    exe = compile_stream_test(str(tmp_path))
    d = str(tmp_path)
    m_ga, m_nga = synth.make_map(10000)
    batch = synth.make_batch(n_scans, n_loop=64)
    for name, a in (("m_ga.f64", m_ga), ("m_nga.f64", m_nga), ("pts.f64", batch.pts), ("scan_off.i32", batch.scan_off),
                    ("scan_nga.i32", batch.scan_nga), ("R0.f64", batch.R), ("t0.f64", batch.t)):
        np.ascontiguousarray(a).tofile(os.path.join(d, name))
    out = os.path.join(d, "out.bin")
    line = subprocess.check_output([exe, d, out, str(chunk), str(size), str(res)]).decode()
    info = json.loads(line.strip().splitlines()[-1])
    assert info["identical"] and info["scans"] == n_scans
    raw = open(out, "rb").read()
    cells = size * size
    pose = np.frombuffer(raw[:16], np.float64)
    o = 16
    R = np.frombuffer(raw[o:o + 32 * n_scans], np.float64).reshape(n_scans, 4); o += 32 * n_scans
    t = np.frombuffer(raw[o:o + 16 * n_scans], np.float64).reshape(n_scans, 2); o += 16 * n_scans
    hits = np.frombuffer(raw[o:o + 4 * cells], np.int32); o += 4 * cells
    misses = np.frombuffer(raw[o:o + 4 * cells], np.int32); o += 4 * cells
    occ = np.frombuffer(raw[o:o + cells], np.int8)

    model = O.IcpModel(m_ga, m_nga)
    Ro, to, iters, ncorr, delta = model.fit_batch(batch.pts, batch.scan_off, batch.scan_nga, batch.R, batch.t,
                                                  O.icp_params(20, 1e-6, 5.0))
    # north-star tolerance (1e-4 m / 1e-5 rad): the sums run in another order, so a query can round to another float
    assert np.abs(t - to).max() < 1e-4 and np.abs(R - Ro).max() < 1e-5
    assert np.abs(t - batch.true_poses[:, :2]).max() < 0.25     # 20 iterations from a 0.3 m / 0.05 rad offset

    gp = O.grid_params(size, size, res, max_range=0.45 * size * res, rolling=1)
    H = np.zeros((size, size), np.int32)
    M = np.zeros((size, size), np.int32)
    num = np.zeros(cells)
    eocc = np.full(cells, -1, np.int8)
    cx = cy = 0.0
    for s0 in range(0, n_scans, chunk):
        s1 = min(s0 + chunk, n_scans)
        dx = int(np.round((batch.t[s0, 0] - cx) / res))     # mls.cpp:419-424 (std::round, half away from zero)
        dy = int(np.round((batch.t[s0, 1] - cy) / res))
        assert abs((batch.t[s0, 0] - cx) / res % 1 - 0.5) > 1e-6
        if dx or dy:
            H, M = roll(H, dx, dy), roll(M, dx, dy)
            cx += dx * res
            cy += dy * res
        for s in range(s0, s1):
            p = batch.pts[batch.scan_off[s]:batch.scan_off[s + 1]]
            Rs = R[s].reshape(2, 2)
            end = ((Rs[0, 0] * p[:, 0] + Rs[0, 1] * p[:, 1] + t[s, 0]) - cx,
                   (Rs[1, 0] * p[:, 0] + Rs[1, 1] * p[:, 1] + t[s, 1]) - cy)
            end = np.stack(end, 1).astype(np.float32)
            org = np.tile(np.array([t[s, 0] - cx, t[s, 1] - cy]).astype(np.float32), (len(p), 1))
            O.grid_raycast(gp, org, end, H.reshape(-1), M.reshape(-1))
    assert (pose[0], pose[1]) == (cx, cy)
    assert np.array_equal(hits, H.reshape(-1)) and np.array_equal(misses, M.reshape(-1))
    assert hits.sum() > 0.5 * batch.n_points and misses.sum() > 20 * hits.sum()
    O.grid_finalize(gp, H.reshape(-1), M.reshape(-1), num, eocc)
    assert np.array_equal(occ, eocc)


@pytest.mark.gpu
@pytest.mark.parametrize("n_scans", [24, 4])
def test_graph_replay_equals_direct_calls(n_scans):
    """One registration + map-update step captured into a hipGraph through the C-ABI and replayed: the same
    poses, counts and occupancy as the calls issued one by one.  24 scans: one workgroup per scan; 4: the spread form
    (a persistent launch + the launch that redoes what it could not finish), whose ordering against other spread launches
    stays outside a captured stream."""
    from slam_amd import api
    m_ga, m_nga = synth.make_map(10000)
    batch = synth.make_batch(n_scans, n_loop=64)
    S, P = batch.n_scans, batch.n_points
    icp = api.Icp(m_ga, m_nga, max_iter=14, min_delta=-1.0)
    grid = api.Grid(1200, 1200, 0.05, rolling=0)
    st = api.Stream()
    d_pts = api.DeviceArray.from_host(batch.pts, np.float64)
    d_off = api.DeviceArray.from_host(batch.scan_off, np.int32)
    d_nga = api.DeviceArray.from_host(batch.scan_nga, np.int32)
    d_R0 = api.DeviceArray.from_host(batch.R, np.float64)
    d_t0 = api.DeviceArray.from_host(batch.t, np.float64)
    d_R = api.DeviceArray(batch.R.shape, np.float64)
    d_t = api.DeviceArray(batch.t.shape, np.float64)
    d_res = api.DeviceArray((S,), api.RESULT_DTYPE)

    def step():
        d_R.copy_from(d_R0, st)
        d_t.copy_from(d_t0, st)
        grid.reset_counts(st)
        icp.fit_batch_dev(d_pts, d_off, d_nga, S, d_R, d_t, 5.0, d_res, None, st)
        grid.raycast_scans_dev(d_pts, d_off, S, P, d_R, d_t, st)
        grid.finalize(st)

    step()                                   # direct (also creates every scratch buffer)
    st.synchronize()
    want = (d_R.download(), d_t.download(), d_res.download(), grid.read_counts(), grid.read_occupancy())
    g = api.Graph(st)
    with g:
        step()
    d_R.zero(); d_t.zero()
    grid.clear()
    api.synchronize()
    for _ in range(3):
        g.launch()
    st.synchronize()
    got = (d_R.download(), d_t.download(), d_res.download(), grid.read_counts(), grid.read_occupancy())
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2]["iters"], want[2]["iters"]) and (got[2]["iters"] == 14).all()
    assert np.array_equal(got[3][0], want[3][0]) and np.array_equal(got[3][1], want[3][1])
    assert np.array_equal(got[4], want[4]) and want[3][0].sum() > 0.9 * P
    icp.close(); grid.close()
