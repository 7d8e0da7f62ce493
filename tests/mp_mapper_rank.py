"""One rank of the two-process rehearsals in test_gpu_two_ranks.py: several processes on GPU 0, each a slam_mapper_t
on its own shard of a scan sequence, merging over the library's host-staged communicator (slam_comm_create_host) with
gloo carrying the host buffers -- so that slam_grid_merge_begin/_finish, the dirty-row bookkeeping, slam_grid_fold and
the accumulator run with more than one rank where RCCL cannot (it does not put two ranks on one device).

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python mp_mapper_rank.py <mode> <out.npz>

mode: fixed (non-rolling grid), rolling (every rank moves its window the same way), apart (rank 1 moves its window
elsewhere: the merge must refuse)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_SCANS, CHUNK, SIZE, RES, MERGE_EVERY = 96, 8, 1000, 0.05, 2


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch                      # torch first: ONE HIP runtime in the process
    import torch.distributed as dist
    from slam_amd import api, synth
    api.set_device(0)                 # every rank on GPU 0
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allreduce(a, op):
        dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM if op == api.COMM_SUM else dist.ReduceOp.MIN)
    comm = api.Comm.host(rank, world, allreduce)
    assert comm.info() == (rank, world)

    m_ga, m_nga = synth.make_map(10000)
    full = synth.make_batch(N_SCANS, n_loop=256)
    mine = full.shard(rank, world)
    rolling = mode != "fixed"
    mp = api.Mapper(m_ga, m_nga, grid=dict(rolling=int(rolling), min_cluster_points=20, max_range=0.45 * SIZE * RES),
                    grid_size_x=SIZE, grid_size_y=SIZE, resolution=RES, max_scans=CHUNK, max_points=CHUNK * 1100,
                    merge_every=MERGE_EVERY, icp=dict(max_iter=20, min_delta=1e-6))
    mp.use_comm(comm)
    n_chunks = mine.n_scans // CHUNK
    R, t = np.zeros((mine.n_scans, 4)), np.zeros((mine.n_scans, 2))
    pending, windows, err = [], [], ""
    try:
        for c in range(n_chunks):
            sub = mine.shard(c, n_chunks)
            # the window position of chunk c is the same on every rank (taken from rank 0's shard) -- except in
            # mode "apart", where rank 1 walks off by itself
            ref = full.shard(0, world).shard(c, n_chunks)
            wx, wy = (float(ref.t[0, 0]), float(ref.t[0, 1])) if rolling else (0.0, 0.0)
            if mode == "apart" and rank == 1:
                wx += 3.0
            windows.append((wx, wy))
            if len(pending) == mp.n_slots:
                slot, a = pending.pop(0)
                R[a:a + CHUNK], t[a:a + CHUNK] = mp.wait(slot)
            pending.append((mp.push(sub, window_xy=(wx, wy)), c * CHUNK))
        for slot, a in pending:
            R[a:a + CHUNK], t[a:a + CHUNK] = mp.wait(slot)
        mp.finish()
    except api.SlamError as ex:
        err = "%d: %s" % (ex.code, ex)
    if err:
        np.savez(out, error=np.array(err))
    else:
        hits, misses = mp.grid.read_counts()
        st = mp.stats()
        np.savez(out, error=np.array(""), R=R, t=t, hits=hits, misses=misses, occ=mp.grid.read_occupancy(),
                 windows=np.array(windows), merges=st["merges"], rows=np.array(st["last_merge_rows"]),
                 pose=np.array(mp.grid.get_pose()), cell=np.array(mp.grid.window_cell()))
    mp.close()
    comm.close()
    if not err:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
