"""The audio FIR of large blocks by overlap-save FFT convolution (csrc/pmr_fir_fft.hip) against the direct MFMA form
(PMR_FIR=direct: exact k-ordered f32 chains = liquid's firfilt order) and against the CPU oracle.

Reference: firfilt_rrrf_execute_block(ctcss_filt) src/sdr_pmr446.c:882, gain :890, de-emphasis :895-899, PCM :903-906; the CTCSS
low-pass branch :884-889.  Same linear filter, different f32 roundings: float audio within 3e-6 of the signal scale of the direct
form (bar: 1e-5), int16 PCM within +-1 LSB of both the direct form and the oracle, a few per cent of the samples at most on the
other side of a truncation boundary."""
import os

import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, active_channels
from sdr_pmr446_amd import synth

pytestmark = pytest.mark.gpu


def _chain(form, **kw):
    """form: None = the product's choice (FFT form on large blocks), "direct", "fft1024", "fft2048", "fft4096" (PMR_FIR, read once at create)."""
    from sdr_pmr446_amd import chain
    old = os.environ.get("PMR_FIR")
    if form:
        os.environ["PMR_FIR"] = form
    try:
        return chain.PmrChain(**kw)
    finally:
        if form:
            if old is None:
                os.environ.pop("PMR_FIR", None)
            else:
                os.environ["PMR_FIR"] = old


def _synth(n, fs, M, **kw):
    """The SURVEY s8(d) stream, ALL M channels, generated on the device (pmr_synth_iq_device) and downloaded."""
    from sdr_pmr446_amd import chain
    buf = chain.synth_iq_device(n, fs, M, **kw)
    x = buf.download(np.complex64, n)
    buf.free()
    return x


def _run(c, x, splits, want):
    outs, pos = {}, 0
    for n in splits:
        o = c.process_block(x[pos:pos + n], want=want)
        pos += n
        for k in want:
            outs.setdefault(k, []).append(o[k])
    return {k: np.concatenate(v, axis=1) for k, v in outs.items()}


CASES = [
    # (fs, M), splits, synthesised channels, open channels (None = all), form under test
    (CFG2, [1 << 21, (1 << 20) + 12345, 786433], None, None, None),             # ragged calls
    (CFG2, [1 << 22], None, [3, 9, 14], None),                                  # open-channel list, odd count (last pair = one channel)
    (CFG2, [(1 << 22) + 999], None, [5], None),                                 # reference semantics: one open channel
    (CFG3, [1 << 23, (1 << 22) + 77], None, None, None),                        # 1706 frames per call
    (CFG3, [1 << 25], None, None, None),                                        # one large call at 256 channels
    (CFG5, [1 << 24, 1 << 24], None, None, None),                               # the headline plan: 209 frames per call, one short block
    # the 4096-point kernels (PMR_FIR=fft4096: 256 threads, 35 KB of LDS; no plan selects them, tests keep them honest)
    (CFG2, [1 << 21, (1 << 20) + 12345, 786433], None, None, "fft4096"),
    (CFG2, [(1 << 22) + 999], None, [5, 12, 13], "fft4096"),
    (CFG3, [1 << 24], None, None, "fft4096"),
    # the 2048-point kernels (PMR_FIR=fft2048: 128 threads, 18 KB of LDS)
    (CFG2, [1 << 21, (1 << 20) + 12345, 786433], None, None, "fft2048"),
    (CFG2, [(1 << 22) + 999], None, [5, 12, 13], "fft2048"),
    (CFG3, [1 << 24], None, None, "fft2048"),
    (CFG5, [1 << 24, 1 << 24], None, None, "fft2048"),
    # ... and the 1024-point kernels forced where the plan now picks 2048 points
    (CFG2, [1 << 21, (1 << 20) + 12345, 786433], None, None, "fft1024"),
    (CFG3, [1 << 24], None, None, "fft1024"),
]
IDS = ["cfg2-ragged", "cfg2-three-open", "cfg2-one-open", "cfg3-two-calls", "cfg3-one-call", "cfg5",
       "cfg2-ragged-4096pt", "cfg2-three-open-4096pt", "cfg3-4096pt", "cfg2-ragged-2048pt", "cfg2-three-open-2048pt", "cfg3-2048pt", "cfg5-2048pt", "cfg2-ragged-1024pt", "cfg3-1024pt"]


@pytest.mark.parametrize("cfg,splits,ks,open_ch,form", CASES, ids=IDS)
def test_fft_form_equals_direct_form_and_oracle(cfg, splits, ks, open_ch, form):
    fs, M = cfg
    n = sum(splits)
    x = _synth(n, fs, M, dev_hz=1500.0)
    mb = max(splits)
    res = []
    for f in (form, "direct"):
        g = _chain(f, fs_in=fs, num_channels=M, max_block=mb)
        if open_ch is not None:
            g.set_channel_mask(open_ch)
        res.append(_run(g, x, splits, ("pcm", "audio")))
        g.close()
    fft, ref = res
    chans = [k for k in active_channels(M, ks, fs) if open_ch is None or k in open_ch]
    assert chans and fft["pcm"].shape == ref["pcm"].shape and fft["pcm"].shape[1] > 200
    scale = float(np.abs(ref["audio"][chans]).max())
    err = float(np.abs(fft["audio"][chans] - ref["audio"][chans]).max())
    assert scale > 0.05 and err <= 3e-6 * max(scale, 1.0), (err, scale)
    d = np.abs(fft["pcm"][chans].astype(np.int32) - ref["pcm"][chans].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.03, (int(d.max()), float((d > 0).mean()))
    if open_ch is not None:                                    # closed rows are untouched by either form
        closed = [k for k in range(M) if k not in open_ch]
        assert np.array_equal(fft["pcm"][closed], ref["pcm"][closed])
    if n <= (1 << 24):                                         # the oracle at 30-50 MS/s: bounded
        o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb)
        po = _run(o, x, splits, ("pcm",))["pcm"]
        o.close()
        assert np.abs(fft["pcm"][chans].astype(np.int32) - po[chans].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("open_ch,form,cfg", [(None, None, CFG2), ([2, 5, 11], None, CFG2), (None, "fft4096", CFG2), ([2, 5, 11], "fft4096", CFG2),
                                              (None, "fft2048", CFG2), ([2, 5, 11], "fft2048", CFG2), (None, "fft1024", CFG2),
                                              ("one", None, CFG2), ("one", None, CFG5)],
                         ids=["all", "three-open", "all-4096pt", "three-open-4096pt", "all-2048pt", "three-open-2048pt", "all-1024pt",
                              "one-open-default-plan", "one-open-default-plan-cfg5"])
def test_fft_form_with_the_ctcss_branch_as_second_product(open_ch, form, cfg):
    """Detector on: the low-pass branch delay188(x) - hp(x) (:884-889) leaves the same forward transform as a second product.
    Branch samples against the direct DUAL pass and the oracle; the detector's events on top of it against the oracle's.
    "one-open-default-plan": the reference's own mode (:893: ONE squelch-opened channel + the detector) on the plan fir_fft_pick selects
    by itself for it (the 2048-point DUAL kernel for <= 2 open channels) -- ADVICE r05: that default had no parity test."""
    fs, M = cfg
    splits = [(1 << 21) + 4321, 1 << 21] if cfg == CFG2 else [(1 << 25) + 4321, 1 << 25]
    if open_ch == "one":
        open_ch = [[k for k in active_channels(M, None, fs) if synth.channel_kind(k) == "fm"][0]]
    x = _synth(sum(splits), fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0)
    res = []
    for f in (form, "direct"):
        g = _chain(f, fs_in=fs, num_channels=M, max_block=max(splits))
        if open_ch is not None:
            g.set_channel_mask(open_ch)
        res.append(_run(g, x, splits, ("pcm", "ctcss_lp", "ctcss")))
        g.close()
    fft, ref = res
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(splits))
    ro = _run(o, x, splits, ("pcm", "ctcss_lp", "ctcss"))
    o.close()
    chans = [k for k in active_channels(M, None, fs) if open_ch is None or k in open_ch]
    scale = float(np.abs(ro["ctcss_lp"][chans]).max())
    assert scale > 0.01
    assert np.abs(fft["ctcss_lp"][chans] - ref["ctcss_lp"][chans]).max() <= 3e-6 * max(scale, 1.0)
    assert np.abs(fft["ctcss_lp"][chans] - ro["ctcss_lp"][chans]).max() <= 1e-5 * max(scale, 1.0)
    assert np.abs(fft["pcm"][chans].astype(np.int32) - ro["pcm"][chans].astype(np.int32)).max() <= 1
    fm_ch = [k for k in chans if synth.channel_kind(k) == "fm"]
    assert fft["ctcss"].shape == ro["ctcss"].shape and (ro["ctcss"].shape[1] >= 8 or cfg != CFG2)
    for k in fm_ch:
        assert np.array_equal(fft["ctcss"]["index"][k], ro["ctcss"]["index"][k])
        assert np.array_equal(fft["ctcss"]["detected"][k], ro["ctcss"]["detected"][k])
        assert np.allclose(fft["ctcss"]["max_power"][k], ro["ctcss"]["max_power"][k], rtol=5e-3)
