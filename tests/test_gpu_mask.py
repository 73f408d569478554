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
    """PCM rows of a chain with the channel mask set, output buffers pre-filled with a sentinel."""
    import torch
    from sdr_pmr446_amd import chain
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(splits), **kw)
    g.set_channel_mask(enabled)
    S = g.max_frames
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x.view(np.float32)).to(dev)
    parts, pos = [], 0
    for n in splits:
        pcm = torch.full((M, S), SENTINEL, dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        ns = g.process_block_device(xd.data_ptr() + pos * 8, n, d_pcm=pcm.data_ptr(), stride=S)
        g.synchronize()
        parts.append(pcm[:, :ns].cpu().numpy())
        pos += n
    g.close()
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
    import torch
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n = 100000
    x = synth.synth_iq(3 * n, fs, M, dev_hz=1500.0)
    kw = dict(lowpass=True, deemph_fir=True)
    full = _gpu_masked(fs, M, x, [n, n, n], None, **kw)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n, **kw)
    S = g.max_frames
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x.view(np.float32)).to(dev)
    out = []
    for b, en in enumerate(([2], [2, 6], [6])):
        g.set_channel_mask(en)
        pcm = torch.full((M, S), SENTINEL, dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        ns = g.process_block_device(xd.data_ptr() + b * n * 8, n, d_pcm=pcm.data_ptr(), stride=S)
        g.synchronize()
        out.append(pcm[:, :ns].cpu().numpy())
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
