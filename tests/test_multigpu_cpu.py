"""N > 1 path on CPU (gloo, world_size 2): one independent IQ stream per rank, no data-path collective; the only
collectives are the barrier around the timed region and the MAX-reduce of the elapsed time (bench.py contract)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    import oracle
    from sdr_pmr446_amd import multigpu, synth
    dist = multigpu.init_dist("gloo")
    assert multigpu.env_world() == (rank, rank, world)
    sid = multigpu.stream_id_for_rank(rank)
    fs, M, n, steps = 2.4e6, 16, 50000, 3
    x = synth.synth_iq(n, fs, M, stream_id=sid, dev_hz=500.0)
    ch = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)     # the checker stands in for the GPU chain here

    def run():
        frames = 0
        for _ in range(steps):
            frames += ch.process_block(x)["n_frames"]
        time.sleep(0.05 * (rank + 1))            # uneven ranks: the reported time must be the slowest one's
        return frames

    t0 = time.perf_counter()
    dt, frames = multigpu.timed_region(run, dist)
    local = time.perf_counter() - t0
    q.put((rank, sid, float(np.abs(x[:64]).sum()), dt, local, frames,
           multigpu.aggregate_throughput(world, steps, n, dt)))
    dist.destroy_process_group()


def test_two_ranks_two_streams_max_time():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, sig0, dt0, loc0, f0, thr0), (r1, s1, sig1, dt1, loc1, f1, thr1) = res
    assert (r0, r1) == (0, 1) and (s0, s1) == (0, 1)
    assert sig0 != sig1                                      # different synthetic streams
    assert dt0 == dt1                                        # MAX-reduced: identical on every rank
    assert dt0 >= 0.1 and dt0 <= max(loc0, loc1) + 0.05     # the slow rank (sleep 0.10 s) sets it
    assert f0 == f1 and thr0 == thr1 == pytest.approx(2 * 3 * 50000 / dt0)


def test_single_process_defaults():
    from sdr_pmr446_amd import multigpu
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert multigpu.env_world() == (0, 0, 1) and multigpu.init_dist("gloo") is None
    dt, out = multigpu.timed_region(lambda: 7)
    assert out == 7 and dt >= 0
    assert multigpu.aggregate_throughput(8, 10, 1 << 20, 2.0) == 8 * 10 * (1 << 20) / 2.0
