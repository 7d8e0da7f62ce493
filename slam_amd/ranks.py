"""Control plane of a multi-rank run (one process per GPU): rendezvous, barriers, the maximum over the ranks' clocks --
and failing FAST.  The data plane is the library's own communicator (include/slam_mi355x_rccl.h); this file never touches
a GPU and is what `bench.py --gpus N` and the multi-rank tests stand on.

Why a watchdog: a rank that dies leaves the others in a collective that never completes (a barrier, an RCCL kernel that
polls a flag the dead peer will never write), until somebody's outer time-out.  Here every rank
  * beats a counter in the job's key-value store (the c10d store of the rendezvous) twice a second from a thread of its own,
  * watches the other ranks' counters and a `failed` key: a rank whose counter has not moved for `dead_after_s`, a rank that
    announced its own failure, or a store that has gone (rank 0 hosts it) ends THIS process with exit code EXIT_PEER_LOST
    and one line on stderr that names the rank -- via os._exit: the main thread may be asleep inside a device wait, and a
    process that has touched the GPU is never re-executed, it just ends;
  * passes a time-out to every barrier (gloo's monitored barrier names the ranks that did not arrive).
"""
import datetime
import os
import sys
import threading
import time

EXIT_PEER_LOST = 3     # another rank (or the store) is gone / said it failed
EXIT_SELF_FAILED = 4   # this rank failed and said so


def _log(msg):
    print(msg, file=sys.stderr, flush=True)


class Ranks:
    def __init__(self, timeout_s=120.0, heartbeat_s=0.5, dead_after_s=15.0, watchdog=True):
        import torch.distributed as dist
        self.dist = dist
        self.timeout = datetime.timedelta(seconds=timeout_s)
        dist.init_process_group("gloo", timeout=self.timeout)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.heartbeat_s, self.dead_after_s = heartbeat_s, dead_after_s
        self._stop = threading.Event()
        self._closing = threading.Event()     # close() was called: keep beating, stop judging the others
        self._thread = None
        self.store = None
        if watchdog and self.world > 1:
            # the rendezvous' own store (torch.distributed.run hands every rank the same one); a prefix keeps the keys apart
            self.store = dist.PrefixStore("slam_ranks", dist.distributed_c10d._get_default_store())
            self.store.set("hb/%d" % self.rank, "0")
            self._thread = threading.Thread(target=self._watch, name="slam-ranks-watchdog", daemon=True)
            self._thread.start()

    # ------------------------------------------------------------------ collectives of the control plane
    def barrier(self, what="barrier"):
        if self.world == 1:
            return
        try:
            self.dist.monitored_barrier(timeout=self.timeout, wait_all_ranks=True)
        except Exception as ex:    # names the ranks that did not arrive (on rank 0), or the lost connection (elsewhere)
            self.fail("%s: %s" % (what, str(ex).splitlines()[0] if str(ex) else repr(ex)))

    def all_reduce(self, tensor, op=None, what="all-reduce"):
        try:
            self.dist.all_reduce(tensor, op=op if op is not None else self.dist.ReduceOp.SUM)
        except Exception as ex:
            self.fail("%s: %s" % (what, str(ex).splitlines()[0] if str(ex) else repr(ex)))
        return tensor

    def max_over_ranks(self, value):
        import torch
        if self.world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        return float(self.all_reduce(t, self.dist.ReduceOp.MAX, "maximum over the ranks")[0].item())

    def broadcast_object(self, obj, src=0):
        box = [obj if self.rank == src else None]
        try:
            self.dist.broadcast_object_list(box, src=src)
        except Exception as ex:
            self.fail("broadcast from rank %d: %s" % (src, str(ex).splitlines()[0] if str(ex) else repr(ex)))
        return box[0]

    # ------------------------------------------------------------------ failing fast
    def fail(self, reason, code=EXIT_SELF_FAILED):
        """This rank cannot go on: tell the others (they exit within a heartbeat), say why, end the process."""
        _log("rank %d of %d FAILED: %s" % (self.rank, self.world, reason))
        try:
            if self.store is not None:
                self.store.set("failed", "%d:%s" % (self.rank, reason[:400]))
        except Exception:
            pass
        sys.stderr.flush()
        os._exit(code)

    def _peer_lost(self, msg):
        _log("rank %d of %d exits: %s" % (self.rank, self.world, msg))
        sys.stderr.flush()
        os._exit(EXIT_PEER_LOST)

    def _watch(self):
        seen = {r: (None, time.monotonic()) for r in range(self.world) if r != self.rank}
        beat = 0
        while not self._stop.wait(self.heartbeat_s):
            beat += 1
            try:
                self.store.set("hb/%d" % self.rank, str(beat))
                if self._closing.is_set():
                    # between close() and the return of its barrier this rank is alive and must look it: a peer that still works
                    # would otherwise see a frozen counter and give up after dead_after_s although the barrier's time-out allows
                    # the longer skew.  It no longer judges the others (rank 0's orderly exit takes the store with it: not a lost
                    # peer); the barrier's own time-out still guards it.
                    continue
                if self.store.check(["failed"]):
                    who, _, why = self.store.get("failed").decode(errors="replace").partition(":")
                    if who != str(self.rank):
                        self._peer_lost("rank %s failed (%s)" % (who, why))
                now = time.monotonic()
                for r, (last, t_last) in list(seen.items()):
                    key = "hb/%d" % r
                    cur = self.store.get(key) if self.store.check([key]) else None
                    if cur != last:
                        seen[r] = (cur, now)
                    elif now - t_last > self.dead_after_s:
                        self._peer_lost("rank %d has stopped responding (no heartbeat for %.0f s)" % (r, now - t_last))
            except Exception as ex:
                if self._stop.is_set() or self._closing.is_set():
                    return
                self._peer_lost("the job's store is gone (%s): rank 0, which hosts it, has ended" % (str(ex).splitlines()[0] if str(ex) else repr(ex)))

    def close(self):
        """Orderly end: stop judging the others (rank 0's exit takes the store with it: not a lost peer) but KEEP BEATING until the
        last barrier has returned -- its time-out still guards it --, then the heartbeat off and the process group down."""
        self._closing.set()
        self.barrier("closing barrier")
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2 * self.heartbeat_s + 1.0)
        try:
            self.dist.destroy_process_group()
        except Exception:
            pass
