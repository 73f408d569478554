"""the waterfall line -- asgramcf of the resampled stream, reference src/sdr_pmr446.c:473-477
(create, set_scale(-40, 2)) and :911-915 (write(resamp_buf, ny), execute, printf).  liquid's asgram/spgram are restated in
oracle/orc_dsp.c; the device computes the averaged periodogram, the host the dB conversion and the character line."""
import numpy as np
import pytest

from parity_util import CFG2, CFG3


def _numpy_psd(x, nfft):
    """float64 model of the stated algorithm: Hann(nfft) scaled as spgram does, 4 nfft bins, every nfft / 2 samples."""
    P = 4 * nfft
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(nfft) / (nfft - 1))
    w = w * (np.sqrt(2) / (np.sqrt((w * w).sum() / nfft) * np.sqrt(P)))
    acc, k = np.zeros(P), 0
    for e in range(nfft // 2, len(x) + 1, nfft // 2):
        seg = np.zeros(nfft, dtype=complex)
        src = x[max(0, e - nfft):e]
        seg[nfft - len(src):] = src
        acc += np.abs(np.fft.fft(seg * w, P)) ** 2
        k += 1
    return 10 * np.log10(np.maximum(np.fft.fftshift(acc), 1e-12) / max(k, 1)), k


def _tone_mix(n, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = 0.3 * np.exp(2j * np.pi * 0.1 * t) + 0.02 * np.exp(-2j * np.pi * 0.31 * t)
    return (x + 0.003 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


@pytest.mark.parametrize("nfft", [16, 64, 256])
def test_oracle_asgram_matches_float64_model(nfft):
    import oracle
    x = _tone_mix(20011)
    a = oracle.OracleAsgram(nfft)
    r = a.block(x)
    ref, k = _numpy_psd(x.astype(np.complex128), nfft)
    assert k == len(x) // (nfft // 2)
    assert np.abs(r["psd_db"] - ref).max() < 2e-3
    assert abs(r["peakfreq"] - 0.1) <= 1.0 / nfft and abs(r["peakval"] - ref.max()) < 2e-3
    # execute resets the periodogram: a second, different block is independent of the first
    r2 = a.block(x[:5000] * 0.5)
    ref2, _ = _numpy_psd(x[:5000].astype(np.complex128) * 0.5, nfft)
    assert np.abs(r2["psd_db"] - ref2).max() < 2e-3
    # fewer than nfft / 2 samples: no transform, a blank line
    r3 = a.block(x[:nfft // 2 - 1])
    assert r3["ascii"] == " " * nfft and r3["peakval"] == 0.0
    a.close()


@pytest.mark.parametrize("nfft", [16, 128])
def test_host_character_line_is_asgram_execute(nfft):
    """pmr_asgram_ascii (library host code) on the oracle's PSD == the oracle's own asgramcf_execute: same characters, same peak."""
    import oracle
    from sdr_pmr446_amd import chain
    a = oracle.OracleAsgram(nfft, ref=-40.0, div=2.0)
    for n, amp in ((30000, 1.0), (7777, 0.01), (nfft // 2 - 1, 1.0)):
        x = _tone_mix(n, seed=n) * amp
        r = a.block(x)
        ntr = n // (nfft // 2)
        line, pv, pf = chain.asgram_ascii(r["psd_db"], nfft, ntr, -40.0, 2.0)
        assert line == r["ascii"]
        assert pv == np.float32(r["peakval"]) and pf == np.float32(r["peakfreq"])
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,nfft,sizes", [(CFG2, 64, [100000, 99999, 1, 250000]), (CFG2, 1024, [1 << 20]),
                                             (CFG3, 128, [1 << 21, 1500001])],
                         ids=["cfg2-64", "cfg2-1024", "cfg3-128"])
def test_device_periodogram_matches_oracle(cfg, nfft, sizes):
    """Every block: PSD of the device's resampled stream vs the oracle's asgram fed the oracle's resampler output; the character
    line may differ only where a group's maximum sits within the PSD tolerance of a level boundary."""
    import oracle
    from sdr_pmr446_amd import chain, synth
    fs, M = cfg
    x = synth.synth_iq(sum(sizes), fs, M, stream_id=11)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
    g.spectrum_enable(nfft)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(sizes))
    a = oracle.OracleAsgram(nfft)
    pos = 0
    for n in sizes:
        blk = x[pos:pos + n]
        pos += n
        g.process_block(blk, want=("pcm",))
        psd, ntr = g.spectrum_read()
        res = o.process_block(blk, want=("pcm", "resampled"))["resampled"]
        r = a.block(res)
        assert ntr == len(res) // (nfft // 2)
        if ntr == 0:
            assert not psd.any()
            continue
        d = np.abs(psd - r["psd_db"])
        assert d.max() < 0.02, "PSD differs by %.4f dB at bin %d" % (d.max(), int(d.argmax()))
        line, pv, pf = chain.asgram_ascii(psd, nfft, ntr)
        assert abs(pv - r["peakval"]) < 0.02
        grp = r["psd_db"].reshape(nfft, 4).max(axis=1)
        levels = -40.0 + 2.0 * np.arange(10)
        near = np.abs(grp[:, None] - levels[None, :]).min(axis=1) < 0.05
        assert all(c1 == c2 or nb for c1, c2, nb in zip(line, r["ascii"], near))
    g.close(); o.close(); a.close()
