"""One rank of the fail-fast tests (tests/test_multi_rank.py): the control plane of bench.py's N > 1 run -- slam_amd.ranks --
on the CPU.  argv: how rank 1 ends after its third round ("kill" = SIGKILL, "stop" = SIGSTOP: alive but silent,
"fail" = it announces a failure of its own, "none" = nobody fails, "late" = nobody fails and rank 0 goes on working for longer than
dead_after_s after rank 1 has reached close()).  RANK / WORLD_SIZE / MASTER_* come from the environment."""
import os
import signal
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.ranks import Ranks  # noqa: E402

how = sys.argv[1]
rk = Ranks(timeout_s=float(os.environ.get("SLAM_RANKS_TIMEOUT", "20")), heartbeat_s=0.25,
           dead_after_s=float(os.environ.get("SLAM_RANKS_DEAD_AFTER", "4")))
for step in range(6):
    rk.barrier("step %d" % step)
    v = rk.max_over_ranks(step + rk.rank)
    assert v == step + rk.world - 1
    if step == 2 and rk.rank == 1:
        if how == "kill":
            os.kill(os.getpid(), signal.SIGKILL)
        if how == "stop":
            os.kill(os.getpid(), signal.SIGSTOP)
        if how == "fail":
            rk.fail("the device reported an error (made up by the test)")
    if step == 2 and rk.rank == 0 and how == "stop":
        time.sleep(3600)       # asleep as in a device wait: only the watchdog can end this process
if how == "late" and rk.rank == 0:
    time.sleep(float(os.environ.get("SLAM_RANKS_DEAD_AFTER", "4")) + 3.0)     # an unbalanced tail: the other rank waits in close()
rk.close()
print("rank %d done" % rk.rank, flush=True)
