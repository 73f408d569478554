"""One independent IQ stream per GPU: process/rank plumbing shared by bench.py and the tests.

The hot path shards by stream: rank r owns stream r, its own handle, its own HBM buffers.  There is NO data-path
collective; torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is used for exactly two
things: the barrier that brackets the timed region and the MAX-reduction of the elapsed time.
"""
import os
import time


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_dist(backend, device=None):
    """Join the process group if WORLD_SIZE > 1.  Returns the torch.distributed module or None."""
    rank, _, world = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def stream_id_for_rank(rank):
    """Rank r processes synthetic stream r (distinct seed: synth.SEED_BASE + stream_id)."""
    return rank


def timed_region(run, dist=None, device_sync=None, reduce_device=None):
    """barrier + sync | run() | sync + barrier; returns the MAX over ranks of the elapsed seconds."""
    import torch
    if dist is not None:
        dist.barrier()
    if device_sync:
        device_sync()
    t0 = time.perf_counter()
    out = run()
    if device_sync:
        device_sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=reduce_device or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


def aggregate_throughput(world, steps, units_per_step, dt_max):
    """Whole-job units/s: every rank processed steps*units_per_step units of ITS OWN stream in dt_max (weak scaling)."""
    return world * steps * units_per_step / dt_max
