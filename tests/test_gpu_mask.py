"""SURVEY s8 row f1, second half: demodulating only the OPEN channels (the reference's own semantics, src/sdr_pmr446.c:876-877)
and the per-channel reset the reference performs when the squelch detunes (:866-867), against the oracle."""
import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, active_channels, pcm_diff
from sdr_pmr446_amd import synth

pytestmark = pytest.mark.gpu

SENTINEL = 12345


def _gpu_masked(fs, M, x, splits, enabled, **kw):
    """PCM rows of a chain with the channel mask set, output buffers pre-filled with a sentinel.  Device buffers come from the
    library's own runtime (include/pmr_mem.h), not from torch."""
    from sdr_pmr446_amd import chain
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(splits), **kw)
    g.set_channel_mask(enabled)
    S = g.max_frames
    xd = chain.DeviceBuffer(x.nbytes); xd.upload(x)
    out = chain.DeviceBuffer(M * S * 2)
    fill = np.full(M * S, SENTINEL, dtype=np.int16)
    parts, pos = [], 0
    for n in splits:
        out.upload(fill)
        ns = g.process_block_device(xd.ptr + pos * 8, n, d_pcm=out.ptr, stride=S)
        g.synchronize()
        parts.append(out.download(np.int16, M * S).reshape(M, S)[:, :ns].copy())
        pos += n
    g.close(); xd.free(); out.free()
    return np.concatenate(parts, axis=1)


@pytest.mark.parametrize("cfg,k,n,splits", [
    (CFG2, 5, 300000, [100000, 1, 99999, 100000]),
    (CFG2, 0, 120000, [120000]),
    (CFG3, 100, 1 << 22, [1 << 21, 1 << 21]),
    (CFG5, 219, 1 << 25, [1 << 24, 1 << 24]),
], ids=["cfg2-ch5", "cfg2-ch0", "cfg3-ch100", "cfg5-ch219"])
def test_one_open_channel_matches_reference_semantics(cfg, k, n, splits):
    """mask = {k}  <->  OracleChain(only_channel=k): the one open channel's PCM within +-1 LSB, nothing else written."""
    fs, M = cfg
    ks = sorted(set([k] + list(range(0, M, max(1, M // 16)))))
    x = synth.synth_iq(n, fs, M, channels=ks, dev_hz=1500.0)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(splits), only_channel=k)
    ref, pos = [], 0
    for s in splits:
        ref.append(o.process_block(x[pos:pos + s])["pcm"]); pos += s
    ref = np.concatenate(ref, axis=1)
    got = _gpu_masked(fs, M, x, splits, [k])
    assert got.shape == ref.shape and got.shape[1] > 100
    assert pcm_diff(got[k], ref[k]).max() <= 1 and np.abs(ref[k]).max() > 1000
    others = [c for c in range(M) if c != k]
    assert np.all(got[others] == SENTINEL)                              # rows of closed channels are left untouched


@pytest.mark.parametrize("cfg,enabled", [(CFG2, [1, 2, 9, 14]), (CFG3, list(range(3, 256, 17)) + [250, 251])],
                         ids=["cfg2", "cfg3"])
def test_open_channels_equal_the_all_channel_run(cfg, enabled):
    """Any set of open channels gives, on those channels, bit for bit what demodulating every channel gives."""
    fs, M = cfg
    n = 200000 if M == 16 else 1 << 21
    x = synth.synth_iq(n, fs, M, channels=enabled, dev_hz=1500.0)
    full = _gpu_masked(fs, M, x, [n // 2, n - n // 2], None)
    part = _gpu_masked(fs, M, x, [n // 2, n - n // 2], enabled)
    assert np.array_equal(part[enabled], full[enabled]) and full.shape[1] > 100
    closed = [c for c in range(M) if c not in enabled]
    assert np.all(part[closed] == SENTINEL)


def test_mask_change_between_blocks_and_lowpass_chain():
    """The mask may change between calls; a channel opened later has current history (the discriminator ran all along), so
    from its first open block on it equals the all-channel run.  Also covers the multi-pass audio chain (deemph FIR + low-pass)."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n = 100000
    x = synth.synth_iq(3 * n, fs, M, dev_hz=1500.0)
    kw = dict(lowpass=True, deemph_fir=True)
    full = _gpu_masked(fs, M, x, [n, n, n], None, **kw)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n, **kw)
    S = g.max_frames
    xd = chain.DeviceBuffer(x.nbytes); xd.upload(x)
    pcm = chain.DeviceBuffer(M * S * 2)
    fill = np.full(M * S, SENTINEL, dtype=np.int16)
    out = []
    for b, en in enumerate(([2], [2, 6], [6])):
        g.set_channel_mask(en)
        pcm.upload(fill)
        ns = g.process_block_device(xd.ptr + b * n * 8, n, d_pcm=pcm.ptr, stride=S)
        g.synchronize()
        out.append(pcm.download(np.int16, M * S).reshape(M, S)[:, :ns].copy())
    f0 = out[0].shape[1]; f1 = f0 + out[1].shape[1]
    assert np.array_equal(out[0][2], full[2, :f0]) and np.array_equal(out[1][2], full[2, f0:f1])
    assert np.array_equal(out[1][6], full[6, f0:f1]) and np.array_equal(out[2][6], full[6, f1:])
    assert np.all(out[2][2] == SENTINEL) and np.all(out[0][6] == SENTINEL)
    g.set_channel_mask(None)


@pytest.mark.parametrize("cfg,k", [(CFG2, 4), (CFG3, 85), (CFG5, 146)], ids=["cfg2", "cfg3", "cfg5"])
def test_reset_channel_is_freqdem_reset(cfg, k):
    """pmr_chain_reset_channel(k) between two blocks == freqdem_reset + ctcss_detector_reset of channel k in the oracle
    (src/sdr_pmr446.c:866-867): that channel's first discriminator output of the next block is exactly 0, everything else as
    without the reset; PCM within +-1 LSB of the oracle."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    n = {16: 100000, 256: 1 << 21, 1024: 1 << 24}[M]
    ks = sorted(set([k] + list(range(0, M, max(1, M // 8)))))
    x = synth.synth_iq(2 * n, fs, M, channels=ks, dev_hz=1500.0)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    o.process_block(x[:n]); g.process_block(x[:n])
    o.reset_channel(k); g.reset_channel(k)
    ro = o.process_block(x[n:], want=("pcm", "fm")); rg = g.process_block(x[n:], want=("pcm", "fm"))
    assert rg["fm"][k, 0] == 0.0 and ro["fm"][k, 0] == 0.0
    assert abs(rg["fm"][k, 1]) > 0
    act = active_channels(M, ks, fs)
    assert np.abs(rg["fm"][act] - ro["fm"][act]).max() < 5e-6
    assert pcm_diff(rg["pcm"][act], ro["pcm"][act]).max() <= 1
    o.close(); g.close()


def test_host_entry_points_leave_closed_channels_rows_untouched():
    """include/pmr_chain.h: 'the pcm / audio rows of disabled channels are left untouched' -- also for the HOST-buffer entry points
    (pmr_chain_process_block_f32, submit / collect, channelize + demodulate), whose outputs pass through compact staging rows that
    hold another block's data for a closed channel.  Rows of open channels equal the all-channel run bit for bit."""
    import ctypes as C
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n, nb = 100000, 4
    x = synth.synth_iq(nb * n, fs, M, dev_hz=1500.0)
    ref = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    full = [ref.process_block(x[b * n:(b + 1) * n], want=("pcm", "audio")) for b in range(nb)]
    ref.close()
    masks = ([3], [3, 9], [9], [0, 15])
    ip = lambda a: a.ctypes.data

    def buffers(S):
        return np.full((M, S), SENTINEL, dtype=np.int16), np.full((M, S), float(SENTINEL), dtype=np.float32)

    def check(b, pcm, audio, ns, en):
        assert ns == full[b]["n_frames"] and ns > 100
        closed = [c for c in range(M) if c not in en]
        assert np.all(pcm[closed] == SENTINEL) and np.all(audio[closed] == float(SENTINEL)), "block %d: closed rows written" % b
        assert np.array_equal(pcm[en, :ns], full[b]["pcm"][en]) and np.array_equal(audio[en, :ns], full[b]["audio"][en])
        assert np.all(pcm[en, ns:] == SENTINEL)

    # synchronous call
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    S = g.max_frames
    for b, en in enumerate(masks):
        g.set_channel_mask(en)
        pcm, audio = buffers(S)
        ns = C.c_uint(0)
        xb = np.ascontiguousarray(x[b * n:(b + 1) * n])
        g._check(g._L.pmr_chain_process_block_f32(g.h, ip(xb), n, ip(pcm), ip(audio), S, C.byref(ns), None, None))
        check(b, pcm, audio, ns.value, en)
    g.close()
    # asynchronous pair: two blocks in flight, the mask changes between the submits
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    keep = []
    for b0 in (0, 2):
        for b in (b0, b0 + 1):
            g.set_channel_mask(masks[b])
            xb = np.ascontiguousarray(x[b * n:(b + 1) * n]); keep.append(xb)
            g._check(g._L.pmr_chain_submit_block(g.h, ip(xb), n, 3))
        for b in (b0, b0 + 1):
            pcm, audio = buffers(S)
            ns = C.c_uint(0)
            g._check(g._L.pmr_chain_collect_block(g.h, ip(pcm), ip(audio), S, C.byref(ns), None, None))
            check(b, pcm, audio, ns.value, masks[b])
    g.close()
    # two-step form: the mask set AFTER the block was channelized is the one that counts
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    for b, en in enumerate(masks):
        xb = np.ascontiguousarray(x[b * n:(b + 1) * n])
        ns = C.c_uint(0)
        rssi = np.zeros(M, dtype=np.float32)
        g._check(g._L.pmr_chain_channelize_block(g.h, ip(xb), n, C.byref(ns), None, 0, ip(rssi)))
        g.set_channel_mask(en)
        pcm, audio = buffers(S)
        g._check(g._L.pmr_chain_demodulate_block(g.h, ip(pcm), ip(audio), S, C.byref(ns)))
        check(b, pcm, audio, ns.value, en)
    g.close()
