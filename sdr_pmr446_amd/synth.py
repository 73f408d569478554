"""Deterministic synthetic multi-channel NBFM IQ -- stands in for the SoapySDR cf32 ingest
(reference src/shared.c:62 SOAPY_SDR_CF32, src/sdr_pmr446.c:789 readStream).

Channel k of M sits at (k - (M-1)/2) * 12.5 kHz from band centre (the layout src/sdr_pmr446.c:25-28,
:432-434 imply: after the NCO shift channel k lands on channelizer bin k).  'fm' channels carry an audio tone
400 + 37*(k mod 64) Hz at 2.5 kHz peak deviation plus the CTCSS tone ctcss_freqs[k mod 38] at 300 Hz
deviation; k % 8 == 3 is a bare carrier; k % 8 == 7 is empty (noise only; excluded from PCM parity because the
discriminator is ill-conditioned there).  Complex AWGN gives 30 dB SNR in each 12.5 kHz channel.
The noise PRNG is counter based (splitmix64 of seed + sample index), so any sub-range is reproducible.
"""
import numpy as np

SEED_BASE = 0x504D523434343600
CHANNEL_WIDTH_HZ = 12500.0

CTCSS_FREQS = np.array([  # standard EIA tone set (same 38 values as reference src/sdr_pmr446.c:138-141)
    67.0, 71.9, 74.4, 77.0, 79.7, 82.5, 85.4, 88.5, 91.5, 94.8, 97.4, 100.0, 103.5, 107.2, 110.9, 114.8, 118.8,
    123.0, 127.3, 131.8, 136.5, 141.3, 146.2, 151.4, 156.7, 162.2, 167.9, 173.8, 179.9, 186.2, 192.8, 203.5,
    210.7, 218.1, 225.7, 233.6, 241.8, 250.3])


def channel_kind(k):
    if k % 8 == 7:
        return "empty"
    if k % 8 == 3:
        return "carrier"
    return "fm"


def dc_block_gain(k, num_channels, fs_in, alpha=0.0005):
    """|H(f_k)| of the chain's own dc blocker (reference src/sdr_pmr446.c:422, H(z) = (1 - z^-1)/(1 - (1-alpha) z^-1)) at the
    centre of channel k.  At GS/s input rates the notch (~alpha fs / 2 pi wide) swallows the channels next to band centre."""
    w = 2.0 * np.pi * (k - (num_channels - 1) / 2.0) * CHANNEL_WIDTH_HZ / fs_in
    z = np.exp(-1j * w)
    return float(abs(1.0 - z) / abs(1.0 - (1.0 - alpha) * z))


def signal_channels(num_channels, fs_in=None, synthesized=None, min_dc_gain=0.5):
    """Channels whose PCM is compared: they carry a signal (not 'empty': the discriminator of pure noise is ill-conditioned,
    SURVEY s7) and, when fs_in is given, the chain's dc-block notch leaves them within 6 dB of nominal (same reason)."""
    ks = range(num_channels) if synthesized is None else synthesized
    return [k for k in ks if channel_kind(k) != "empty" and
            (fs_in is None or dc_block_gain(k, num_channels, fs_in) >= min_dc_gain)]


def audio_tone_hz(k):
    return 400.0 + 37.0 * (k % 64)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _uniform01(seed, idx):
    with np.errstate(over="ignore"):
        z = _splitmix64(np.uint64(seed) + idx)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def awgn(seed, n0, n, sigma):
    """Complex Gaussian noise, total variance sigma^2, for absolute sample indices n0 .. n0+n-1."""
    idx = (np.arange(n0, n0 + n, dtype=np.uint64)) * np.uint64(2)
    u1 = _uniform01(seed, idx)
    u2 = _uniform01(seed, idx + np.uint64(1))
    r = np.sqrt(-np.log(u1)) * sigma          # sqrt(-2 ln u) * sigma/sqrt(2)
    return r * np.exp(2j * np.pi * u2)


def synth_iq(n, fs_in, num_channels, stream_id=0, n0=0, snr_db=30.0, channels=None, dc_offset=0.0,
             chunk=1 << 18, dev_hz=2500.0, ctcss_dev_hz=300.0, ctcss_dev_of=None):
    """Return complex64 IQ samples [n0, n0+n) of stream `stream_id`.

    channels: iterable of channel indices to synthesise (default: all M).  Amplitudes/noise do not depend on it.
    ctcss_dev_of: optional callable k -> CTCSS deviation in Hz of channel k (default: ctcss_dev_hz for every channel).
    """
    M = num_channels
    seed = (SEED_BASE + stream_id) & 0xFFFFFFFFFFFFFFFF
    amp = 0.5 / np.sqrt(M)
    sigma = np.sqrt(amp * amp / (10.0 ** (snr_db / 10.0)) * (fs_in / CHANNEL_WIDTH_HZ))
    ks = list(range(M)) if channels is None else list(channels)
    out = np.empty(n, dtype=np.complex64)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        t = (np.arange(n0 + c0, n0 + c1, dtype=np.float64)) / fs_in
        acc = awgn(seed, n0 + c0, c1 - c0, sigma) + dc_offset
        for k in ks:
            kind = channel_kind(k)
            if kind == "empty":
                continue
            fk = (k - (M - 1) / 2.0) * CHANNEL_WIDTH_HZ
            ph0 = 2.0 * np.pi * _uniform01(seed ^ 0xA5A5A5A5, np.array([k], dtype=np.uint64))[0]
            ph = 2.0 * np.pi * fk * t + ph0
            if kind == "fm":
                fa = audio_tone_hz(k)
                fc = CTCSS_FREQS[k % 38]
                cd = ctcss_dev_hz if ctcss_dev_of is None else float(ctcss_dev_of(k))
                ph = ph + (dev_hz / fa) * np.sin(2.0 * np.pi * fa * t) + (cd / fc) * np.sin(2.0 * np.pi * fc * t)
            acc = acc + amp * np.exp(1j * ph)
        out[c0:c1] = acc.astype(np.complex64)
    return out
