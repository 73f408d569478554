"""The code path bench.py TIMES, checked against the oracle: consecutive pmr_chain_process_block_device() calls with NO
synchronisation in between, block pipelining on (front end of block b+1 under the back end of block b, three blocks in
flight, ring reuse gated by events), at the bench's block size (2^26 samples) on cfg2, cfg3 and cfg5.

Stands for the reference's block loop, src/sdr_pmr446.c:788-908 (one readStream block per iteration, state carried).
Bar: concatenated int16 PCM within +-1 LSB of the CPU oracle on every channel that carries a signal, and BIT-IDENTICAL to
the same blocks run with pipelining off (every block start-to-finish before the next)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from parity_util import CFG2, CFG3, CFG5, active_channels

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_pcm(fs, M, x_host, chunk=1 << 22, **kw):
    """PCM of the whole stream from the CPU oracle (fed in `chunk`-sample blocks: the oracle is bit-exact under re-blocking,
    tests/test_oracle_model.py)."""
    import oracle
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=chunk, **kw)
    parts = [o.process_block(x_host[p:p + chunk], want=("pcm",))["pcm"] for p in range(0, len(x_host), chunk)]
    o.close()
    return np.concatenate(parts, axis=1)


def run_device_blocks(g, iq, sizes, sync_each=False, **outs):
    """Feed consecutive blocks of a device-resident stream; returns the concatenated PCM (torch int16 [M, frames])."""
    import torch
    M, S = g.M, g.max_frames
    pcm = torch.zeros((len(sizes), M, S), dtype=torch.int16, device=iq.device)      # one output buffer per block
    torch.cuda.synchronize()                       # torch filled iq / pcm on ITS stream; the chain's streams are non-blocking
    ns, pos = [], 0
    for b, n in enumerate(sizes):
        ns.append(g.process_block_device(iq.data_ptr() + pos * 8, n, d_pcm=pcm[b].data_ptr(), stride=S))
        pos += n
        if sync_each:
            g.synchronize()
    g.synchronize()
    return torch.cat([pcm[b, :, :ns[b]] for b in range(len(sizes))], dim=1), ns


@pytest.mark.parametrize("cfg,step", [(CFG2, 1), (CFG3, 5), (CFG5, 73)], ids=["cfg2", "cfg3", "cfg5"])
def test_bench_path_unsynchronised_full_size_blocks(cfg, step):
    import torch
    from sdr_pmr446_amd import chain
    from sdr_pmr446_amd.synth_torch import synth_iq_torch
    fs, M = cfg
    block, nblk = 1 << 26, 7                       # 2 * PIPE_DEPTH + 1 blocks: every ring slot / event is reused twice
    ks = list(range(0, M, step))
    dev = torch.device("cuda", 0)
    iq = synth_iq_torch(nblk * block, fs, M, dev, dev_hz=1500.0, channels=ks)       # ONE stream, 7 distinct consecutive blocks
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
    piped, ns = run_device_blocks(g, iq, [block] * nblk)
    g.reset()
    g.set_overlap(False)
    serial, ns2 = run_device_blocks(g, iq, [block] * nblk)
    g.close()
    assert ns == ns2 and torch.equal(piped, serial), "pipelined blocks differ from the same blocks run one at a time"
    ref = oracle_pcm(fs, M, iq.cpu().numpy())
    got = piped.cpu().numpy()
    assert got.shape == ref.shape and got.shape[1] > 5000
    act = active_channels(M, ks)
    d = np.abs(got[act].astype(np.int32) - ref[act].astype(np.int32))
    assert d.max() <= 1, "PCM differs from the oracle by %d LSB (frame %d)" % (d.max(), int(np.argmax(d.max(axis=0))))
    assert np.abs(ref[act]).max() > 1000


CASE = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, torch
from sdr_pmr446_amd import chain
from sdr_pmr446_amd.synth_torch import synth_iq_torch
from test_gpu_pipelined import run_device_blocks, oracle_pcm
from parity_util import active_channels
fs, M, sizes, ctcss = %(fs)r, %(M)r, %(sizes)r, %(ctcss)r
ks = list(range(M)) if M <= 64 else list(range(0, M, M // 16))
iq = synth_iq_torch(sum(sizes), fs, M, torch.device("cuda", 0), dev_hz=1500.0, channels=ks)
g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
if ctcss:
    g._check(g._L.pmr_chain_ctcss_enable(g.h, 1))
piped, ns = run_device_blocks(g, iq, sizes)
ev1 = g.ctcss_read() if ctcss else None
g.reset()
serial, ns2 = run_device_blocks(g, iq, sizes, sync_each=True)
ev2 = g.ctcss_read() if ctcss else None
assert ns == ns2, (ns, ns2)
assert torch.equal(piped, serial), "un-synchronised pipelined calls differ from per-block-synchronised calls"
if ctcss:
    assert ev1.shape == ev2.shape and np.array_equal(ev1["index"], ev2["index"]) and np.array_equal(ev1["max_power"], ev2["max_power"])
ref = oracle_pcm(fs, M, iq.cpu().numpy())
got = piped.cpu().numpy()
act = active_channels(M, ks)
d = int(np.abs(got[act].astype(np.int32) - ref[act].astype(np.int32)).max())
print("frames", got.shape[1], "maxdiff", d)
sys.exit(0 if (got.shape == ref.shape and d <= 1 and got.shape[1] > 50) else 1)
"""

RAGGED2 = [1 << 21, 1500001, 7, 0, 2000000, 123457, 1 << 21, 999999]
RAGGED5 = [1 << 24, 9000001, 4097, 0, 1 << 23, 12345679, 1 << 24]
RAGGED3 = [1 << 22, 3000001, 4097, 0, 1 << 22, 2345679, 1 << 22]

CASES = [
    ({}, CFG2, RAGGED2, False),
    ({}, CFG3, RAGGED3, False),
    ({}, CFG5, RAGGED5, False),
    ({}, CFG2, RAGGED2, True),                         # CTCSS branch enabled (second FIR pass + detector kernels in the back end)
    ({}, CFG3, RAGGED3, True),
    ({"PMR_FIR": "direct"}, CFG2, RAGGED2, False),     # the direct MFMA form of the audio FIR on the big blocks (default: FFT form)
    ({"PMR_CARRY": "inplace"}, CFG3, RAGGED3, False),  # dc carry by the in-place pass (what debug capture / waterfall calls fall back to)
]


@pytest.mark.parametrize("env,cfg,sizes,ctcss", CASES,
                         ids=["%s-%dch%s" % ("+".join("%s=%s" % kv for kv in e.items()) or "default", c[1], "-ctcss" if ct else "")
                              for e, c, _, ct in CASES])
def test_ragged_blocks_in_flight_match_synchronised_run(env, cfg, sizes, ctcss):
    """>= 2 * PIPE_DEPTH + 1 consecutive ragged blocks, distinct output buffers, nothing synchronised in between: bit-identical
    to the per-block-synchronised run and within +-1 LSB of the oracle.  One process per case (switches are read once)."""
    e = dict(os.environ)
    e.update(env)
    src = CASE % dict(root=ROOT, fs=cfg[0], M=cfg[1], sizes=sizes, ctcss=ctcss)
    r = subprocess.run([sys.executable, "-c", src], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])


RESET_CASE = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, torch
from sdr_pmr446_amd import chain
from sdr_pmr446_amd.synth_torch import synth_iq_torch
from test_gpu_pipelined import run_device_blocks
fs, M, sizes = %(fs)r, %(M)r, %(sizes)r
iq = synth_iq_torch(sum(sizes), fs, M, torch.device("cuda", 0), dev_hz=1500.0, channels=list(range(M)) if M <= 64 else list(range(0, M, M // 16)))
g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
fresh, ns = run_device_blocks(g, iq, sizes)
g.reset()
S = g.max_frames
scratch = torch.zeros((3, M, S), dtype=torch.int16, device=iq.device)
torch.cuda.synchronize()
# three blocks of a DIFFERENT part of the stream queued and NOT synchronised, then reset() straight away
for b in range(3):
    g.process_block_device(iq.data_ptr() + (sizes[0] // 2) * 8, sizes[0] // 2 + b, d_pcm=scratch[b].data_ptr(), stride=S)
g.reset()
again, ns2 = run_device_blocks(g, iq, sizes)
g.close()
ok = ns == ns2 and torch.equal(fresh, again)
print("frames", fresh.shape[1], "identical", bool(ok))
sys.exit(0 if ok and fresh.shape[1] > 50 else 1)
"""

RESET_CASES = [({}, CFG2, RAGGED2), ({"PMR_CARRY": "inplace"}, CFG2, RAGGED2), ({}, CFG5, RAGGED5)]


@pytest.mark.parametrize("env,cfg,sizes", RESET_CASES,
                         ids=["%s-%dch" % ("+".join("%s=%s" % kv for kv in e.items()) or "default", c[1]) for e, c, _ in RESET_CASES])
def test_reset_with_blocks_in_flight_restarts_cleanly(env, cfg, sizes):
    """pmr_chain_reset() right after un-synchronised device calls (their work still queued on both streams): the restarted
    stream is bit-identical to a fresh handle's -- no zeroing overtaken by in-flight kernels (staged front end: the dc state and
    half-band histories live on the front-end stream)."""
    e = dict(os.environ)
    e.update(env)
    src = RESET_CASE % dict(root=ROOT, fs=cfg[0], M=cfg[1], sizes=[s for s in sizes if s])
    r = subprocess.run([sys.executable, "-c", src], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
