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


class _stdout_to_stderr:
    """gloo announces its connections on the C++ stdout; the bench contract is ONE JSON line there.  While the process groups come up,
    file descriptor 1 points at stderr."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Dist:
    """What bench.py needs from torch.distributed: a barrier and the MAX of one number -- on the backend that actually came up."""

    def __init__(self, dist, group, backend_used, device, note=None):
        self.dist, self.group, self.backend_used, self.device, self.note = dist, group, backend_used, device, note

    def barrier(self):
        if self.backend_used == "nccl":                      # an all-reduce on this rank's device + a device sync: no device guessing
            import torch
            t = torch.zeros(1, dtype=torch.float32, device=self.device)
            self.dist.all_reduce(t, group=self.group)
            torch.cuda.synchronize(self.device)
        else:
            self.dist.barrier(group=self.group)

    def max_float(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=self.device if self.backend_used == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def sum_rows_cpu(self, t):
        """SUM all-reduce of a CPU tensor over the WORLD group (always gloo: see init_dist) -- rank r fills row r, zeros elsewhere."""
        self.dist.all_reduce(t)
        return t

    def destroy(self):
        self.dist.destroy_process_group()


def init_dist(backend, device=None, nccl_timeout_s=120):
    """Join the process group if WORLD_SIZE > 1.  Returns a Dist or None.

    The world group is always gloo (it only carries a barrier and one float64): that cannot fail for GPU reasons.  With
    backend "nccl" (= RCCL on ROCm) an RCCL group is created on top and proven with one all-reduce; the ranks then AGREE (MIN over
    gloo) whether it works everywhere -- if it does the timed region's barrier / MAX run on it, if not they stay on gloo and the
    record says so (`backend_used`), instead of the first multi-GPU run dying in communicator set-up."""
    rank, _, world = env_world()
    if world <= 1:
        return None
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    with _stdout_to_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.zeros(1, dtype=torch.int32)
        dist.all_reduce(t)                                   # connects the pairs now, while stdout is still diverted
    if backend != "nccl":
        return Dist(dist, None, "gloo", None)
    ok, note, grp = 1, None, None
    try:
        # (RCCL prints its version banner on the C stdout when the first communicator comes up -- rank 0's stdout carries ONE JSON
        #  line: profiles/r04_bench_2rank_nccl_fallback.json of round 4 had five banner lines in front of it)
        with _stdout_to_stderr():
            grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=nccl_timeout_s))
            t = torch.ones(1, dtype=torch.float64, device=device)
            dist.all_reduce(t, group=grp)
            torch.cuda.synchronize(device)
        if int(t.item()) != world:
            raise RuntimeError("all_reduce returned %r" % t.item())
    except Exception as e:                                   # reported in the record, never swallowed
        ok, note = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)             # gloo: every rank learns whether RCCL came up EVERYWHERE
    if int(flag.item()) == 1:
        return Dist(dist, grp, "nccl", device)
    return Dist(dist, None, "gloo", None, note or "RCCL group failed on another rank")


def reduce_parity(dist, rank, world, rec):
    """Every rank's oracle verdict on ITS stream and device -> one record every rank holds: {ranks, all_ok, worst_lsb, per_rank}.
    `rec`: the rank's own parity record (bench.parity_check: ok, max_abs_pcm_diff_lsb, frames_checked, channels_checked, device).
    The exchange is one SUM all-reduce of an int64 [world][6] table over the gloo world group (rank r fills row r).  The last column:
    how many PCM samples of the rank's check differ from the oracle by more than 1 LSB inside the ill-conditioned classes of
    parity_rule.py (samples outside them at > 1 LSB make `ok` false)."""
    lsb = rec.get("max_abs_pcm_diff_lsb")
    over = (rec.get("ill_conditioned") or {}).get("samples_over_1_lsb")
    row = [1 if rec.get("ok") else 0, int(lsb) if isinstance(lsb, int) and lsb >= 0 else -1, int(rec.get("frames_checked") or 0),
           int(rec.get("channels_checked") or 0), int(rec.get("device", -1)), int(over) if isinstance(over, int) else -1]
    if dist is None or world <= 1:
        rows = [row]
    else:
        import torch
        t = torch.zeros((world, 6), dtype=torch.int64)
        t[rank] = torch.tensor(row, dtype=torch.int64)
        rows = dist.sum_rows_cpu(t).tolist()
    bad = [r for r in rows if r[0] != 1]
    return {"ranks": len(rows), "all_ok": not bad,
            "worst_lsb": None if any(r[1] < 0 for r in rows) else max(r[1] for r in rows),
            "per_rank": [{"rank": i, "device": r[4], "ok": bool(r[0]), "max_abs_pcm_diff_lsb": r[1] if r[1] >= 0 else None,
                          "frames_checked": r[2], "channels_checked": r[3],
                          **({"ill_conditioned_samples_over_1_lsb": r[5]} if r[5] >= 0 else {})} for i, r in enumerate(rows)]}


def bind_to_gpu_numa(local_rank, world_local):
    """Best effort: pin this rank's host threads to the cores next to its GPU (the host-fed legs: H2D staging, the oracle check).
    Uses the GPU's PCI function in sysfs (local_cpulist); falls back to an even split of the visible cores.  Returns a note."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        return "affinity unsupported"
    want = None
    try:
        import torch
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            spec = f.read().strip()
        want = set()
        for part in spec.split(","):
            if "-" in part:
                lo, hi = part.split("-"); want.update(range(int(lo), int(hi) + 1))
            elif part:
                want.add(int(part))
        want = sorted(want & set(cores))
        how = "cores local to GPU %s" % bdf
    except Exception:
        want = None
    if not want:
        n = max(1, len(cores) // max(1, world_local))
        want = cores[(local_rank % max(1, world_local)) * n:(local_rank % max(1, world_local) + 1) * n] or cores
        how = "even split of the visible cores"
    try:
        os.sched_setaffinity(0, want)
    except Exception as e:
        return "sched_setaffinity failed: %s" % e
    return "%d %s" % (len(want), how)


def self_launch(script, argv, nproc, port=None, python=None, extra_env=None):
    """`python script --gpus N` without an external launcher: start `python -m torch.distributed.run --nproc-per-node N script argv`
    as a CHILD process (this process has made no GPU call and makes none), relay its stdout / stderr and return its exit code.
    Never an exec: a process image that initialised the GPU must not be replaced, and the parent stays alive to report."""
    import socket
    import subprocess
    import sys
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this pool's driver
    env.update(extra_env or {})
    p = subprocess.Popen(cmd, env=env)
    try:
        return p.wait()
    except KeyboardInterrupt:
        p.terminate()
        return p.wait()


def stream_id_for_rank(rank):
    """Rank r processes synthetic stream r (distinct seed: synth.SEED_BASE + stream_id)."""
    return rank


def timed_region(run, dist=None, device_sync=None, reduce_device=None):
    """barrier + sync | run() | sync + barrier; returns the MAX over ranks of the elapsed seconds.  `dist`: a Dist or None."""
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
        dt = dist.max_float(dt)
    return dt, out


def aggregate_throughput(world, steps, units_per_step, dt_max):
    """Whole-job units/s: every rank processed steps*units_per_step units of ITS OWN stream in dt_max (weak scaling)."""
    return world * steps * units_per_step / dt_max
