"""The one-level front end's dc carry applied WHERE THE CHANNELIZER LOADS the resampled stream (pmr_carry_fix, pmr_kernels.h):
reference stages src/sdr_pmr446.c:795 (iirfilt_crcf dc blocker) feeding :796 and :804-814.

The front end's tiles run the blocker from zero state; the missing carry is subtracted later.  Two forms produce it:
  * PMR_CARRY=inplace -- k_fe_tilefix rewrites the whole resampled block (a 2 x 8 x rate B/sample read-modify-write);
  * default           -- the carry pass corrects only the block's tail in place (what later calls re-read as history) and the
                         16- / 256-channel channelizers subtract the same term from every other sample as they load it.
Bar: the two forms give BIT-IDENTICAL PCM on ragged, un-synchronised, pipelined blocks (incl. blocks shorter than the
in-place tail and empty ones), and that PCM is within +-1 LSB of the CPU oracle on every channel that carries a signal.
Buffers come from the library's own runtime (include/pmr_mem.h): no torch in this file."""
import os

import numpy as np
import pytest

from parity_util import CFG2, CFG3, CFG_REF, active_channels

pytestmark = pytest.mark.gpu


def run_stream(cfg, sizes, env, sync_each=False):
    """PCM [M, frames] of one device-resident synthetic stream fed in `sizes` blocks through the device entry point."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))      # switches are read once, here
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    at_load = g.info(chain.INFO_CARRY_AT_LOAD)
    iq = chain.synth_iq_device(sum(sizes), fs, M, dev_hz=1500.0)
    S = g.max_frames
    out = chain.DeviceBuffer(len(sizes) * M * S * 2)
    chain.device_synchronize()
    ns, pos = [], 0
    for b, n in enumerate(sizes):
        ns.append(g.process_block_device(iq.ptr + pos * 8, n, d_pcm=out.ptr + b * M * S * 2, stride=S))
        pos += n
        if sync_each:
            g.synchronize()
    g.synchronize()
    parts = [out.download(np.int16, M * S, b * M * S * 2).reshape(M, S)[:, :ns[b]] for b in range(len(sizes))]
    x = iq.download(np.complex64, sum(sizes))
    g.close(); iq.free(); out.free()
    return np.concatenate(parts, axis=1), ns, at_load, x


RAGGED = {
    "ref": (CFG_REF, [100000, 1, 99999, 0, 3000, 100000, 17, 65536, 100000]),
    "cfg2": (CFG2, [1 << 22, 1500001, 7, 0, 4000, 2000000, 123457, 1 << 21, 999]),
    "cfg3": (CFG3, [1 << 22, 3000001, 4097, 0, 150000, 1 << 21, 2345679, 1 << 22, 70000]),
}


@pytest.mark.parametrize("name", ["ref", "cfg2", "cfg3"])
def test_carry_at_load_equals_in_place_and_oracle(name):
    import oracle
    cfg, sizes = RAGGED[name]
    fs, M = cfg
    a, ns_a, at_load, x = run_stream(cfg, sizes, {})
    b, ns_b, inplace, _ = run_stream(cfg, sizes, {"PMR_CARRY": "inplace"})
    assert at_load == 1 and inplace == 0, "the default plan must apply the carry at load here (%d, %d)" % (at_load, inplace)
    assert ns_a == ns_b and a.shape == b.shape and a.shape[1] > 500
    assert np.array_equal(a, b), "carry at load differs from carry in place: %d samples, max %d LSB" % (
        int((a != b).sum()), int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max()))
    c, ns_c, _, _ = run_stream(cfg, sizes, {}, sync_each=True)
    assert ns_c == ns_a and np.array_equal(a, c), "un-synchronised pipelined calls differ from synchronised ones"
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(sizes))
    ref = np.concatenate([o.process_block(x[p:p + n], want=("pcm",))["pcm"]
                          for p, n in zip(np.cumsum([0] + sizes[:-1]), sizes)], axis=1)
    o.close()
    act = active_channels(M, None, fs)
    d = np.abs(a[act].astype(np.int32) - ref[act].astype(np.int32))
    assert ref.shape == a.shape and d.max() <= 1, "PCM differs from the oracle by %d LSB" % d.max()
    assert np.abs(ref[act]).max() > 1000


def test_debug_capture_and_waterfall_fall_back_to_in_place():
    """Whatever else reads the resampled ring (debug tap-off, asgramcf) must see corrected samples: those calls take the in-place
    form, and switching between the forms in mid-stream changes nothing."""
    from sdr_pmr446_amd import chain, synth
    fs, M = CFG2
    n = 300000
    x = synth.synth_iq(3 * n, fs, M, dev_hz=1500.0)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    h = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    assert g.info(chain.INFO_CARRY_AT_LOAD) == 1
    h._check(h._L.pmr_chain_debug_enable(h.h, 1))
    for b in range(3):
        if b == 2:
            h._check(h._L.pmr_chain_debug_enable(h.h, 0))      # back to the at-load form for the last block
        pg = g.process_block(x[b * n:(b + 1) * n], want=("pcm",))["pcm"]
        ph = h.process_block(x[b * n:(b + 1) * n], want=("pcm",))["pcm"]
        assert np.array_equal(pg, ph), "block %d" % b
    g.close(); h.close()
