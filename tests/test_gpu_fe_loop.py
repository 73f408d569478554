"""Streaming form of the specialised front end (k_fe_stream, csrc/pmr_fe_fast.hip): a workgroup walks several consecutive tiles and
requests tile n + 1 while tile n's dc scan + cascade run.  Reference stages: src/sdr_pmr446.c:795 (iirfilt_crcf_execute_block) and
:796 (msresamp_crcf_execute).

Bar: the same statements per sample as k_fe_fast, so the PCM of a stream is BIT-IDENTICAL between the two forms -- on ragged,
un-synchronised, pipelined blocks whose tile counts do not divide by the run length, with edge tiles (history in front, zeros
behind, blocks that start at odd addresses) inside a run -- and within +-1 LSB of the CPU oracle."""
import os

import numpy as np
import pytest

from parity_util import CFG2, CFG3, CFG5, active_channels

pytestmark = pytest.mark.gpu

INFO_FE_TPW = 11


def run_stream(cfg, sizes, env, offset=0):
    """PCM [M, frames] of one device-resident synthetic stream fed in `sizes` blocks through the device entry point; the tiles per
    workgroup each call's front end used."""
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
    iq = chain.synth_iq_device(sum(sizes) + offset, fs, M, dev_hz=1500.0)
    S = g.max_frames
    out = chain.DeviceBuffer(len(sizes) * M * S * 2)
    chain.device_synchronize()
    ns, tpw, pos = [], [], offset
    for b, n in enumerate(sizes):
        ns.append(g.process_block_device(iq.ptr + pos * 8, n, d_pcm=out.ptr + b * M * S * 2, stride=S))
        tpw.append(g.info(INFO_FE_TPW))
        pos += n
    g.synchronize()
    parts = [out.download(np.int16, M * S, b * M * S * 2).reshape(M, S)[:, :ns[b]] for b in range(len(sizes))]
    x = iq.download(np.complex64, sum(sizes) + offset)[offset:]
    g.close(); iq.free(); out.free()
    return np.concatenate(parts, axis=1), ns, tpw, x


# blocks of >= 4096 tiles take the streaming form; the others (and the empty one) the one-tile form, in the same stream
CASES = {
    "cfg2": (CFG2, [1 << 24, 20000001, 7, 0, 1 << 25, 17000003, 4000]),
    "cfg3": (CFG3, [1 << 24, 18000001, 4097, 0, (1 << 24) + 12345, 70000]),
    "cfg5": (CFG5, [1 << 25, 17000001, 0, (1 << 24) + 777, 1 << 22]),
}


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg5"])
@pytest.mark.parametrize("tpw", [8, 3])
def test_streaming_front_end_is_bit_identical(name, tpw):
    cfg, sizes = CASES[name]
    a, ns_a, tpw_a, _ = run_stream(cfg, sizes, {"PMR_FE_TPW": str(tpw)})
    b, ns_b, tpw_b, _ = run_stream(cfg, sizes, {"PMR_FE_TPW": "0"})
    assert all(t == 0 for t in tpw_b)
    assert [t for t, n in zip(tpw_a, sizes) if n >= (1 << 24)] == [tpw] * sum(n >= (1 << 24) for n in sizes), tpw_a
    assert [t for t, n in zip(tpw_a, sizes) if n < (1 << 22)] == [0] * sum(n < (1 << 22) for n in sizes), tpw_a
    assert ns_a == ns_b and a.shape == b.shape and a.shape[1] > 500
    assert np.array_equal(a, b), "streaming form differs from one tile per workgroup: %d samples, max %d LSB" % (
        int((a != b).sum()), int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max()))


def test_streaming_front_end_unaligned_block_and_oracle():
    """A block that starts 8 bytes off a 16-byte boundary takes the plain-load path for EVERY tile of every run; and the streaming
    form against the oracle."""
    import oracle
    fs, M = CFG2
    sizes = [(1 << 24) + 1, 1 << 24]
    a, ns_a, tpw_a, x = run_stream(CFG2, sizes, {"PMR_FE_TPW": "8"}, offset=1)
    b, ns_b, _, _ = run_stream(CFG2, sizes, {"PMR_FE_TPW": "0"}, offset=1)
    assert tpw_a == [8, 8] and ns_a == ns_b
    assert np.array_equal(a, b)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(sizes))
    ref = np.concatenate([o.process_block(x[p:p + n], want=("pcm",))["pcm"]
                          for p, n in zip(np.cumsum([0] + sizes[:-1]), sizes)], axis=1)
    o.close()
    act = active_channels(M, None, fs)
    d = np.abs(a[act].astype(np.int32) - ref[act].astype(np.int32))
    assert ref.shape == a.shape and d.max() <= 1, "PCM differs from the oracle by %d LSB" % d.max()
    assert np.abs(ref[act]).max() > 1000
