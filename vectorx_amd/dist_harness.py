"""Process-per-GPU timing harness shared by bench.py and the CPU (gloo) tests.

Proof-level sharding: every rank runs the same `step` on its own unit of work (its own witness); there is no
data-path collective — `torch.distributed` is used for the barrier that brackets the timed region and for the
MAX-over-ranks of the elapsed time.  Whole-job throughput = world * steps / max_time ("weak" scaling).
"""
from __future__ import annotations

import os
import time


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, local_rank: int):
    """Returns the torch.distributed module when WORLD_SIZE > 1, else None."""
    _, world, _ = env_rank()
    if world <= 1 and not os.environ.get("VX_FORCE_DIST"):   # VX_FORCE_DIST=1: exercise the collective path at world 1
        return None
    import datetime

    import torch
    import torch.distributed as dist
    # an explicit collective timeout (the first exchange of a run is where a missing peer shows): VX_DIST_TIMEOUT_S, default 10 minutes
    timeout = datetime.timedelta(seconds=float(os.environ.get("VX_DIST_TIMEOUT_S", "600")))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)
    else:
        dist.init_process_group(backend, timeout=timeout)
    return dist


def run_timed(step, steps: int, warmup: int, sync, dist=None, device=None, before_timed=None):
    """W untimed steps, then exactly K timed steps bracketed by sync+barrier on both sides.
    Returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    if before_timed is not None:
        before_timed()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate(world: int, steps: int, dt: float):
    return {"value": world * steps / dt, "ms_per_step": dt / steps * 1e3}
