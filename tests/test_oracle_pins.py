"""Pin the CPU oracle to every numeric fact the reference itself fixes and to independent
numpy/scipy re-derivations.  The reference has no tests or golden vectors (parity unpinned, oracle/README.md);
these are the known answers that exist."""
import hashlib
import math
import os

import numpy as np
import pytest
from scipy import signal

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAPS = np.load(os.path.join(GOLD, "pmr446_taps.npz"))


def _sha(a):
    return hashlib.sha256(np.asarray(a, dtype="<f4").tobytes()).hexdigest()[:16]


def test_tap_tables_match_reference_hashes():
    # sha256 prefixes of reference src/sdr_pmr446.c:56-141 recorded in SURVEY.md s4
    assert _sha(TAPS["hp_audio_taps"]) == "e87946af1ae2f775" and len(TAPS["hp_audio_taps"]) == 377
    assert _sha(TAPS["lp_audio_taps"]) == "62a037b97474b915" and len(TAPS["lp_audio_taps"]) == 103
    assert _sha(TAPS["deemph_taps"]) == "6b0013ceaeaa4d7e" and len(TAPS["deemph_taps"]) == 101
    assert _sha(TAPS["ctcss_freqs"]) == "afa362a349568fb8" and len(TAPS["ctcss_freqs"]) == 38


def test_tap_header_equals_fixture():
    hdr = open(os.path.join(os.path.dirname(GOLD), "..", "sdr_pmr446_amd", "data", "pmr446_taps.h")).read()
    import re
    for name in ("hp_audio_taps", "lp_audio_taps", "deemph_taps", "ctcss_freqs"):
        body = re.search(r"pmr446_%s\[[^\]]*\] = \{(.*?)\};" % name, hdr, re.S).group(1)
        vals = np.array([np.float32(float.fromhex(v.rstrip("f"))) for v in re.findall(r"-?0x[0-9a-fp.+-]+f", body)],
                        dtype=np.float32)
        assert np.array_equal(vals, TAPS[name])


def test_tap_tables_are_the_documented_filters():
    hp, lp = TAPS["hp_audio_taps"].astype(float), TAPS["lp_audio_taps"].astype(float)
    assert np.allclose(hp, hp[::-1]) and np.allclose(lp, lp[::-1])            # linear phase
    f = np.array([100.0, 250.0, 300.0, 400.0, 1000.0, 3000.0])
    _, H = signal.freqz(hp, worN=2 * np.pi * f / 12500.0)
    db = 20 * np.log10(np.abs(H))
    assert db[0] < -77 and db[1] < -77 and db[2] < -77                        # CTCSS band rejected >= 77 dB
    assert abs(db[3]) < 0.3 and abs(db[4]) < 0.1 and abs(db[5]) < 0.1         # voice band flat
    _, H = signal.freqz(lp, worN=2 * np.pi * np.array([1000.0, 4500.0, 5000.0]) / 12500.0)
    db = 20 * np.log10(np.abs(H))
    assert abs(db[0]) < 0.2 and abs(db[1]) < 0.3 and db[2] < -60


# (the de-emphasis coefficients: tests/test_ref_fixtures.py, against what the reference's scripts/filter_des.py returns)


def test_buffer_sizing_asserts_of_the_reference():
    # src/sdr_pmr446.c:730-736: res_size == 39064 and chan_size == 2441 at the reference's operating point
    c = oracle.OracleChain(fs_in=1024000.0, num_channels=16, max_block=100000)
    assert c.max_resampled == 39064 and c.max_frames == 2441


def test_nco_offset_and_constrain():
    off = np.float32(-0.5) * np.float32(15) / np.float32(16) * 2 * np.pi       # :432-433
    assert np.float32(off) == np.float32(-2.9452431)
    assert oracle.lib().orc_nco_constrain(float(np.float32(off))) == 0x88000000  # -15/32 cycle == +17/32 mod 1


def test_msresamp_structure():
    for fs, M, h, ms, step in ((1.024e6, 16, 2, [10, 5], 21474836), (2.4e6, 16, 3, [10, 5, 3], 25165824),
                               (61.44e6, 256, 4, [10, 5, 3, 3], 20132660), (1e9, 1024, 6, [10, 5, 3, 3, 3, 3], 20480000)):
        d = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=1000).design_dict()
        assert d["num_stages"] == h and d["m_stage"] == ms and d["arb_step"] == step


def test_kaiser_design_vs_numpy():
    # firdes_kaiser(n, fc, As) == sinc(2 fc t) * kaiser(beta) -- SURVEY A.1; beta(80) = 7.8573, beta(60) = 5.6533
    assert oracle.lib().orc_kaiser_beta_As(80.0) == pytest.approx(7.85726, abs=1e-4)
    assert oracle.lib().orc_kaiser_beta_As(60.0) == pytest.approx(5.65326, abs=1e-4)
    for n, fc, As in ((417, 0.5 / 16, 80.0), (6657, 0.5 / 256, 80.0), (3585, 0.4 / 256, 60.0)):
        h = oracle.firdes_kaiser(n, fc, As)
        t = np.arange(n) - (n - 1) / 2.0
        ref = np.sinc(2 * fc * t) * np.kaiser(n, float(oracle.lib().orc_kaiser_beta_As(As)))
        assert np.abs(h - ref).max() < 2e-7


def test_channelizer_prototype_shape():
    # sum(h) ~= M, -6 dB at the channel edge +-0.5/M, < -80 dB one channel away
    for M in (16, 256):
        h = oracle.firdes_kaiser(2 * M * 13 + 1, 0.5 / M, 80.0).astype(float)
        assert abs(h.sum() - M) < 0.01 * M
        _, H = signal.freqz(h, worN=2 * np.pi * np.array([0.0, 0.5 / M, 1.0 / M]))
        db = 20 * np.log10(np.abs(H) / np.abs(H[0]))
        assert abs(db[1] + 6.02) < 0.1 and db[2] < -80


def test_pcm_rule():
    f = oracle.lib().orc_pcm_from_float
    assert f(0.0) == 0 and f(0.5) == 16383 and f(-0.5) == -16383          # truncation toward zero (dsd_in.c:174)
    assert f(1.0) == 32767 and f(1.6) == 32767 and f(-1.6) == -32768      # saturation (build addition)
    assert f(0.99999) == 32766 and f(float("nan")) == 0


def test_fft_is_a_forward_dft():
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for n in (2, 16, 256, 1024):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        y = np.zeros(n, dtype=np.complex64)
        f = L.orc_fft_create(n)
        L.orc_fft_forward(f, x.ctypes.data, y.ctypes.data)
        L.orc_fft_destroy(f)
        assert np.abs(y - np.fft.fft(x.astype(np.complex128))).max() < 1e-5 * np.sqrt(n) * 4
