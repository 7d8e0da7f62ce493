"""The N > 1 path run by LIBRARY code with two ranks on the one GPU of the box (BASELINE configs 4 / 5):
two processes on GPU 0, each a slam_mapper_t on its shard, merging through slam_grid_merge_begin / _finish /
slam_grid_fold over the library's host-staged communicator (slam_comm_create_host; gloo carries the host buffers --
RCCL does not put two ranks on one device).  Merged planes == the oracle's Bresenham of the UNION of the ranks' scans,
bit for bit; occupancy == the oracle's finalize of those counts."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import synth
from oracle_lib import roll

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(mode, tmp_path, world=2):
    port = _free_port()
    outs = [str(tmp_path / ("%s_rank%d.npz" % (mode, r))) for r in range(world)]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mp_mapper_rank.py"), mode, outs[r]], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        try:
            _, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-3000:]
    return [np.load(o) for o in outs]


def union_oracle(res, rolling):
    import mp_mapper_rank as W
    size, cell = W.SIZE, W.RES
    full = synth.make_batch(W.N_SCANS, n_loop=256)
    world = len(res)
    gp = O.grid_params(size, size, cell, max_range=0.45 * size * cell, rolling=int(rolling), min_cluster_points=20)
    H, M = np.zeros((size, size), np.int32), np.zeros((size, size), np.int32)
    cx = cy = 0.0
    n_chunks = W.N_SCANS // world // W.CHUNK
    for c in range(n_chunks):
        if rolling:
            wx, wy = res[0]["windows"][c]
            dx, dy = int(np.round((wx - cx) / cell)), int(np.round((wy - cy) / cell))     # mls.cpp:419-424
            if dx or dy:
                H, M = roll(H, dx, dy), roll(M, dx, dy)
                cx += dx * cell
                cy += dy * cell
        for r in range(world):
            mine = full.shard(r, world)
            for s in range(c * W.CHUNK, (c + 1) * W.CHUNK):
                p = mine.pts[mine.scan_off[s]:mine.scan_off[s + 1]]
                Rs, ts = res[r]["R"][s].reshape(2, 2), res[r]["t"][s]
                end = np.stack([(Rs[0, 0] * p[:, 0] + Rs[0, 1] * p[:, 1] + ts[0]) - cx,
                                (Rs[1, 0] * p[:, 0] + Rs[1, 1] * p[:, 1] + ts[1]) - cy], 1).astype(np.float32)
                org = np.tile(np.array([ts[0] - cx, ts[1] - cy]).astype(np.float32), (len(p), 1))
                O.grid_raycast(gp, org, end, H.reshape(-1), M.reshape(-1))
    num, occ = np.zeros(size * size), np.full(size * size, -1, np.int8)
    O.grid_finalize(gp, H.reshape(-1), M.reshape(-1), num, occ)
    return H.reshape(-1), M.reshape(-1), occ, (cx, cy)


@pytest.mark.parametrize("mode,world", [("fixed", 2), ("rolling", 2), ("fixed", 3)])
def test_mapper_two_ranks_rehearsal_on_one_gpu(mode, world, tmp_path):
    import mp_mapper_rank as W
    res = run_ranks(mode, tmp_path, world)
    for r in res:
        assert str(r["error"]) == "", str(r["error"])
    a, b = res[0], res[-1]
    # every rank ends with the merged map
    for other in res[1:]:
        assert np.array_equal(a["hits"], other["hits"]) and np.array_equal(a["misses"], other["misses"]) and np.array_equal(a["occ"], other["occ"])
    # each rank registered its own scans: against the oracle on the same target
    m_ga, m_nga = synth.make_map(10000)
    full = synth.make_batch(W.N_SCANS, n_loop=256)
    model = O.IcpModel(m_ga, m_nga)
    for k, r in enumerate(res):
        mine = full.shard(k, world)
        Ro, to, _, _, _ = model.fit_batch(mine.pts, mine.scan_off, mine.scan_nga, mine.R, mine.t, O.icp_params(20, 1e-6, 5.0))
        assert np.abs(r["t"] - to).max() < 1e-4 and np.abs(r["R"] - Ro).max() < 1e-5
    H, M, occ, pose = union_oracle(res, mode == "rolling")
    assert H.sum() > 0
    assert np.array_equal(a["hits"], H) and np.array_equal(a["misses"], M)      # both ranks' updates, each exactly once
    assert np.array_equal(a["occ"], occ)
    n_chunks = W.N_SCANS // world // W.CHUNK
    assert int(a["merges"]) == n_chunks // W.MERGE_EVERY + 1                     # every second chunk, and once more at finish
    if mode == "rolling":
        assert tuple(a["pose"]) == pose and tuple(b["pose"]) == pose
        assert tuple(a["cell"]) == tuple(b["cell"]) and tuple(a["cell"]) != (0, 0)


def test_merge_refuses_windows_that_moved_apart(tmp_path):
    """slam_grid_merge_finish: storage rows mean the same world cells only while the ranks' rolling windows sit on the same
    cells; ranks whose windows differ get SLAM_E_INVALID (all of them: nobody is left waiting in a collective)."""
    res = run_ranks("apart", tmp_path)
    for r in res:
        assert str(r["error"]).startswith("-1:") and "different cells" in str(r["error"]), str(r["error"])
