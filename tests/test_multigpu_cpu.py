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
    assert dist.backend_used == "gloo"
    dist.destroy()


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


def _parity_worker(rank, world, port, q, fail_rank):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import oracle
    from sdr_pmr446_amd import multigpu, synth
    dist = multigpu.init_dist("gloo")
    fs, M, n = 2.4e6, 16, 60000
    x = synth.synth_iq(n, fs, M, stream_id=multigpu.stream_id_for_rank(rank), dev_hz=500.0)
    ref = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n).process_block(x)["pcm"].astype(np.int32)
    got = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n).process_block(x)["pcm"].astype(np.int32)   # stands in for the GPU chain
    if rank == fail_rank:
        got[3, 100] += 5                                     # this rank's "device" computed something else
    d = int(np.abs(got - ref).max())
    rec = {"ok": d <= 1, "max_abs_pcm_diff_lsb": d, "frames_checked": int(got.shape[1]), "channels_checked": M, "device": rank}
    q.put((rank, multigpu.reduce_parity(dist, rank, world, rec)))
    dist.destroy()


@pytest.mark.parametrize("fail_rank", [-1, 1], ids=["all-ranks-agree", "rank-1-fails"])
def test_every_rank_learns_every_ranks_parity_verdict(fail_rank):
    """bench.py at world > 1: each rank checks ITS stream on ITS device against the oracle and the verdicts are
    reduced over the gloo group -- one failing rank must fail the record on EVERY rank (and with it the job's exit code)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_parity_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                  # the same record on both ranks
    r = res[0]
    assert r["ranks"] == 2 and [e["rank"] for e in r["per_rank"]] == [0, 1] and [e["device"] for e in r["per_rank"]] == [0, 1]
    assert all(e["frames_checked"] > 200 and e["channels_checked"] == 16 for e in r["per_rank"])
    if fail_rank < 0:
        assert r["all_ok"] and r["worst_lsb"] == 0 and all(e["ok"] for e in r["per_rank"])
    else:
        assert not r["all_ok"] and r["worst_lsb"] == 5
        assert [e["ok"] for e in r["per_rank"]] == [True, False] and r["per_rank"][1]["max_abs_pcm_diff_lsb"] == 5


def test_reduce_parity_single_process():
    from sdr_pmr446_amd import multigpu
    r = multigpu.reduce_parity(None, 0, 1, {"ok": True, "max_abs_pcm_diff_lsb": 1, "frames_checked": 10, "channels_checked": 4, "device": 0})
    assert r == {"ranks": 1, "all_ok": True, "worst_lsb": 1,
                 "per_rank": [{"rank": 0, "device": 0, "ok": True, "max_abs_pcm_diff_lsb": 1, "frames_checked": 10, "channels_checked": 4}]}
    r = multigpu.reduce_parity(None, 0, 1, {"ok": False, "error": "oracle died"})
    assert not r["all_ok"] and r["worst_lsb"] is None


def test_single_process_defaults():
    from sdr_pmr446_amd import multigpu
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert multigpu.env_world() == (0, 0, 1) and multigpu.init_dist("gloo") is None
    dt, out = multigpu.timed_region(lambda: 7)
    assert out == 7 and dt >= 0
    assert multigpu.aggregate_throughput(8, 10, 1 << 20, 2.0) == 8 * 10 * (1 << 20) / 2.0


_RANK_SCRIPT = """
import json, os, sys
sys.path.insert(0, %r)
from sdr_pmr446_amd import multigpu
rank, local_rank, world = multigpu.env_world()
dist = multigpu.init_dist("gloo")
dt, out = multigpu.timed_region(lambda: rank + 1, dist)
if rank == 0:
    print(json.dumps({"world": world, "backend_used": dist.backend_used, "args": sys.argv[1:], "out": out}), flush=True)
dist.destroy()
sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == rank and 3 or 0)
"""


def test_self_launch_starts_the_ranks_relays_output_and_exit_code(tmp_path):
    """`python bench.py --gpus N` without an external launcher: the parent spawns torch.distributed.run as a CHILD,
    rank 0's JSON line arrives on the parent's stdout, a failing rank's exit code is not lost."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % root)
    parent = ("import sys; sys.path.insert(0, %r)\nfrom sdr_pmr446_amd import multigpu\n"
              "sys.exit(multigpu.self_launch(%r, ['--x', '1'], 2))" % (root, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", parent], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    assert [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("{")] == [], r.stdout     # gloo's chatter went to stderr
    rec = json.loads(line[0])
    assert rec == {"world": 2, "backend_used": "gloo", "args": ["--x", "1"], "out": 1}
    r = subprocess.run([sys.executable, "-c", parent], env=dict(env, FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_bench_gpus_2_launches_itself():
    """bench.py --gpus 2 with no WORLD_SIZE set must not ask for a launcher: it starts one.  Here (no GPU) the ranks stop at
    'needs a GPU' -- which proves the children ran bench.py with the same arguments -- and the parent relays the failure."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by profiles/r05_bench_2rank_1gpu.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "launch with torch.distributed.run" not in r.stderr
