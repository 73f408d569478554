"""The sequential oracle against the closed-form index sums of tests/chain_model.py (independent float64
restatement with numpy/scipy), against physics-level known answers, and against itself under re-blocking."""
import os

import numpy as np
import pytest

import chain_model as cm
import oracle
from parity_util import CFG2, CFG3, CFG_REF, active_channels, run_blocks
from sdr_pmr446_amd import synth

TAPS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pmr446_taps.npz"))


def _rel(a, b):
    n = min(a.shape[-1], b.shape[-1])
    return np.abs(a[..., :n] - b[..., :n]).max() / np.abs(b[..., :n]).max()


@pytest.mark.parametrize("fs,M,N", [CFG_REF + (60000,), CFG2 + (100000,), CFG3 + (400000,)])
def test_oracle_equals_closed_form_model(fs, M, N):
    ks = list(range(0, M, max(1, M // 16)))
    x = synth.synth_iq(N, fs, M, channels=ks, dev_hz=500.0)
    c = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N)
    o = c.process_block(x, want=("pcm", "chan", "resampled", "fm", "audio", "ctcss_lp"))
    m = cm.run_model(x, c.design_dict(), M, TAPS["hp_audio_taps"])
    ns = o["n_frames"]
    assert len(o["resampled"]) == len(m["resampled"]) and m["chan"].shape[1] == ns
    assert _rel(o["resampled"], m["resampled"]) < 1e-5
    assert _rel(o["chan"], m["chan"]) < 1e-5
    act = active_channels(M, ks)
    assert np.abs(o["fm"][act] - m["fm"][act][:, :ns]).max() < 2e-5
    assert np.abs(o["audio"][act] - m["audio"][act][:, :ns]).max() < 5e-5
    assert np.abs(o["ctcss_lp"][act] - m["ctcss_lp"][act][:, :ns]).max() < 2e-5
    d = np.abs(cm.pcm_from_float(m["audio"][act][:, :ns]).astype(int) - o["pcm"][act].astype(int))
    assert d.max() <= 1


@pytest.mark.parametrize("opts", [dict(lowpass=True), dict(deemph_fir=True), dict(lowpass=True, deemph_fir=True)])
def test_oracle_audio_options_equal_model(opts):
    fs, M, N = CFG2 + (150000,)
    x = synth.synth_iq(N, fs, M, dev_hz=500.0)
    c = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N, **opts)
    o = c.process_block(x, want=("audio",))
    m = cm.run_model(x, c.design_dict(), M, TAPS["hp_audio_taps"], lowpass=opts.get("lowpass", False),
                     lp=TAPS["lp_audio_taps"], deemph_fir=opts.get("deemph_fir", False), deemph_taps=TAPS["deemph_taps"])
    act = active_channels(M)
    assert np.abs(o["audio"][act] - m["audio"][act][:, :o["n_frames"]]).max() < 5e-5


def test_tone_lands_in_its_channel_bin():
    # channel k (carrier at (k - (M-1)/2) * 12.5 kHz) must appear at channelizer index k: src/sdr_pmr446.c:25-28,
    # :432-434, :819-821, :838-839 ("Tuned to channel active_chan+1")
    fs, M, N = CFG2 + (60000,)
    for k in (0, 5, 9, 14):
        t = np.arange(N) / fs
        x = (0.3 * np.exp(2j * np.pi * (k - (M - 1) / 2.0) * 12500.0 * t)).astype(np.complex64)
        c = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N)
        o = c.process_block(x, want=("rssi", "chan"))
        assert int(np.argmax(o["rssi"])) == k
        p = 20 * np.log10(np.abs(o["chan"][:, 60:]).mean(axis=1) + 1e-30)   # skip the switch-on transient
        assert p[k] - np.delete(p, k).max() > 60.0   # >= 60 dB to every other channel (As = 60/80 dB designs)


def test_fm_deviation_gives_known_discriminator_amplitude():
    # kf = 0.5 => output = delta_phi / pi = 2 * dev / fs_channel: 1 kHz deviation -> 0.16
    fs, M, N, k, dev, fa = 2.4e6, 16, 200000, 6, 1000.0, 700.0
    t = np.arange(N) / fs
    ph = 2 * np.pi * (k - 7.5) * 12500.0 * t + (dev / fa) * np.sin(2 * np.pi * fa * t)
    x = (0.2 * np.exp(1j * ph)).astype(np.complex64)
    c = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N)
    o = c.process_block(x, want=("fm", "audio"))
    fm = o["fm"][k, 100:]
    assert abs(np.abs(fm).max() - 2 * dev / 12500.0) < 0.002
    # audio = gain 4 * HP (flat at 700 Hz) * de-emphasis(700 Hz): |H| = 1/sqrt(1 + (f/3183)^2) (bilinear-warped)
    a = o["audio"][k, 500:]
    expect = 4 * 0.16 / np.sqrt(1 + (fa / 3183.1) ** 2)
    assert abs(np.abs(a).max() - expect) < 0.01


def test_ctcss_tone_is_removed_from_audio():
    fs, M, N, k = 2.4e6, 16, 300000, 4
    t = np.arange(N) / fs
    ph = 2 * np.pi * (k - 7.5) * 12500.0 * t + (300.0 / 67.0) * np.sin(2 * np.pi * 67.0 * t)
    x = (0.2 * np.exp(1j * ph)).astype(np.complex64)
    c = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N)
    o = c.process_block(x, want=("fm", "audio", "ctcss_lp"))
    assert np.abs(o["fm"][k, 500:]).max() > 0.04                  # 2*300/12500 = 0.048 before the high-pass
    assert np.abs(o["audio"][k, 800:]).max() < 4 * 0.048 * 10 ** (-70 / 20.0)
    assert np.abs(o["ctcss_lp"][k, 800:]).max() > 0.04            # the complementary branch keeps the tone


@pytest.mark.parametrize("fs,M", [CFG_REF, CFG2])
def test_oracle_block_split_invariance_is_bit_exact(fs, M):
    # src/sdr_pmr446.c:797-823: the 0..M-1 sample remainder carries over, block boundaries are invisible
    N = 250000
    x = synth.synth_iq(N, fs, M, dev_hz=500.0)
    one = run_blocks(oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N), x, [N], ("pcm", "chan"))
    rng = np.random.default_rng(7)
    sp, left = [], N
    while left:
        n = int(min(left, rng.integers(0, 40000)))
        sp.append(n); left -= n
    many = run_blocks(oracle.OracleChain(fs_in=fs, num_channels=M, max_block=40000), x, sp, ("pcm", "chan"))
    assert one["n_frames"] == many["n_frames"]
    assert np.array_equal(one["pcm"], many["pcm"]) and np.array_equal(one["chan"], many["chan"])


def test_only_channel_restores_reference_semantics():
    fs, M, N = CFG_REF + (100000,)
    x = synth.synth_iq(N, fs, M, dev_hz=500.0)
    a = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N).process_block(x)
    b = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=N, only_channel=5).process_block(x)
    assert np.array_equal(a["pcm"][5], b["pcm"][5]) and not b["pcm"][4].any()
