"""How int16 PCM of the HIP chain is held against the CPU oracle's when the comparison starts at a RESET (bench.py's `parity_checked`,
tests/test_gpu_streams.py).  Checker-side helper: pure numpy on arrays both sides produced; it computes nothing of the chain.

The bar is +-1 LSB (BASELINE.json north_star).  One class of samples cannot meet it for a reason that is not the kernels': the
discriminator (freqdem, reference src/sdr_pmr446.c:881) is arg(conj(r') r), and arg() of a near-zero product is ill-conditioned.
Right after a reset the polyphase windows still hold pre-stream zeros: a channel's output climbs from nothing to its steady level
within the bank's 26 frames, its first samples are sums that nearly cancel, and the two implementations' f32 rounding of them -- far
inside the 1e-5 bar for float intermediates -- can turn the PHASE by anything up to pi (round 6, stream 5 at cfg3: PCM 37979 LSB apart at the audio
filter's centre lag behind such a sample; profiles/r06_stream_parity.txt).  The audio FIR (377-tap high-pass x gain x de-emphasis, :882-898) carries such a discriminator difference into the PCM of
the next ~400 frames.  Round 5 applied a blanket window (the first 26 + 383 frames of EVERY channel, <= 8 LSB): both too lax (it hid
every channel's start-up) and too strict (the case above).  This rule is tied to the cause, with MEASURED quantities only:

  ill(k, t)   min(|r(k, t)|, |r(k, t-1)|) < COND_FRAC * rms_k     r = the ORACLE's channelizer output, rms_k its steady level
              or conj(r(k, t-1)) r(k, t) on the branch cut of arg(): Re < 0, |Im| < CUT_FRAC |.|   (+pi and -pi are one angle, 2 apart)
  D(k, t)     fm_chain(k, t) - fm_oracle(k, t) where ill(k, t), 0 elsewhere        (both discriminator outputs, debug taps)
  E(k, t)     32767 * sum_n h[n] D(k, t - n)                  h = the audio path's impulse response (high-pass * gain -> de-emphasis)
  verdict     |pcm_chain - pcm_oracle| <= 1                   wherever E = 0   (every sample no ill-conditioned input reaches)
              |pcm_chain - pcm_oracle - E| <= 1 + min(1, |E|) elsewhere (up to one LSB more: two truncations), unless either side saturates

i.e. the PCM may differ from the oracle's by exactly what the discriminator difference AT ILL-CONDITIONED SAMPLES explains through
the (linear) audio filter, and by the ordinary +-1 otherwise; a discriminator difference at a well-conditioned sample explains
nothing.  The record counts the ill samples, the PCM samples their response reaches, and how many of those actually differ by more
than 1 LSB.  A stream never restarts in the reference (:788), and the reference demodulates only the squelch-opened channel
(:834-836, :876-881), whose |r| is far above the noise by construction: the class exists only because this chain demodulates every
channel from sample 0.
"""
import numpy as np

COND_FRAC = 0.01            # discriminator inputs below 1 % of the channel's steady-state rms are ill-conditioned
PFB_FRAMES = 26             # frames until the polyphase windows hold stream samples only (p = 2 m, src/sdr_pmr446.c:437)
E_FLOOR = 0.02              # |E| below this many LSB counts as "explains nothing" (the +-1 bar applies)
CUT_FRAC = 1e-4             # |Im| below this fraction of |conj(r') r| with Re < 0: on the branch cut of arg() (f32 rounding of r: ~1e-6)


def audio_response(hp_taps, gain, b0, b1, a1, tail=1e-9, deemph_fir_taps=None, lp_taps=None):
    """Impulse response of the audio path behind the discriminator (:882-902): high-pass FIR, gain, de-emphasis -- the IIR
    (v0 = u - a1 v1; y = b0 v0 + b1 v1; truncated where its tail falls below `tail` of the peak) or, with `deemph_fir_taps`, the
    101-tap FIR variant (:457-458) --, then the optional low-pass FIR (`lp_taps`, :453-454).  float64."""
    hp = np.asarray(hp_taps, np.float64) * float(gain)
    if deemph_fir_taps is not None:
        y = np.convolve(hp, np.asarray(deemph_fir_taps, np.float64))
    else:
        extra = 1
        while abs(a1) ** extra > tail and extra < 64:
            extra += 1
        u = np.concatenate([hp, np.zeros(extra)])
        y = np.zeros_like(u)
        v1 = 0.0
        for i, x in enumerate(u):
            v0 = x - a1 * v1
            y[i] = b0 * v0 + b1 * v1
            v1 = v0
    if lp_taps is not None:
        y = np.convolve(y, np.asarray(lp_taps, np.float64))
    return y


def ill_conditioned(chan_ref):
    """chan_ref: complex [K, T], the oracle's channelizer outputs of the compared channels from the reset on -> bool [K, T].
    Two ways for arg(conj(r') r) to be ill-conditioned: an input that is (numerically) nothing, or a product on the BRANCH CUT of
    arg() -- a phase advance of pi per frame, where the sign of a rounding-sized imaginary part decides between +pi and -pi
    (discriminator outputs 2 apart for the same angle).  The second kind needs energy at exactly half the channel rate from the
    channel's centre: the dc blocker's start-up transient does that to the two channels next to band centre (an even channel count
    puts DC on their common edge)."""
    c = np.asarray(chan_ref).astype(np.complex128)
    mag = np.abs(c)
    K, T = mag.shape
    steady = mag[:, PFB_FRAMES:] if T > 2 * PFB_FRAMES else mag
    rms = np.sqrt((steady ** 2).mean(axis=1))
    prevc = np.concatenate([np.zeros((K, 1), c.dtype), c[:, :-1]], axis=1)           # r' of the first frame is the reset state, 0
    small = np.minimum(mag, np.abs(prevc)) < COND_FRAC * rms[:, None]
    z = np.conj(prevc) * c
    cut = (z.real < 0) & (np.abs(z.imag) < CUT_FRAC * np.abs(z))
    return small | cut


def check(got, ref, chan_ref, fm_got, fm_ref, h):
    """got / ref: int [K, T] PCM of the compared channels (HIP chain / oracle) from the reset on; chan_ref: complex [K, T] (oracle);
    fm_got / fm_ref: float [K, F] discriminator outputs of both sides for the first F <= T frames (F must reach past the last
    ill-conditioned frame); h: audio_response().  Returns the verdict record."""
    got = np.asarray(got, np.int64); ref = np.asarray(ref, np.int64)
    chan_ref = np.asarray(chan_ref)
    if got.shape != ref.shape or got.shape != chan_ref.shape:
        return {"ok": False, "error": "shape mismatch %r / %r / %r" % (got.shape, ref.shape, chan_ref.shape)}
    K, T = got.shape
    ill = ill_conditioned(chan_ref)
    last = int(np.nonzero(ill.any(axis=0))[0].max()) if ill.any() else -1
    fm_got = np.asarray(fm_got, np.float64); fm_ref = np.asarray(fm_ref, np.float64)
    F = fm_got.shape[1]
    if fm_got.shape != fm_ref.shape or fm_got.shape[0] != K or F > T:
        return {"ok": False, "error": "discriminator taps %r / %r do not fit %d channels x %d frames" % (fm_got.shape, fm_ref.shape, K, T)}
    # (ill-conditioned samples beyond the taps' F frames have no measured difference: they explain nothing, the +-1 bar applies to
    #  whatever they reach -- the start-up, where both kinds occur, is what the callers tap)
    D = np.where(ill[:, :F], fm_got - fm_ref, 0.0)
    E = np.zeros((K, T))
    h = np.asarray(h, np.float64)
    for k in np.nonzero(np.abs(D).max(axis=1) > 0)[0] if F else []:
        e = np.convolve(D[k], h)[:T]
        E[k, :len(e)] = 32767.0 * e
    d = got - ref
    reached = np.abs(E) > E_FLOOR
    sat = (np.abs(got) >= 32767) | (np.abs(ref) >= 32767)
    strict_bad = (np.abs(d) > 1) & ~reached
    expl_bad = (np.abs(d - E) > 1.0 + np.minimum(1.0, np.abs(E))) & reached & ~sat      # one LSB more only where E itself is >= 1
    w_strict = int(np.abs(d[~reached]).max()) if (~reached).any() else 0
    over = (np.abs(d) > 1) & reached
    return {"ok": bool(not strict_bad.any() and not expl_bad.any()),
            "max_abs_pcm_diff_lsb": w_strict, "tolerance_lsb": 1,
            "ill_conditioned": {
                "rule": "discriminator input below %g of the channel's steady rms (oracle chan, either of the two samples), or conj(r') r on the "
                        "branch cut of arg() (+pi / -pi) = ill-conditioned; "
                        "PCM may differ by what the MEASURED discriminator difference at those samples explains through the audio filter "
                        "(|d - E| <= 1 + min(1, |E|) LSB), by +-1 LSB everywhere else" % COND_FRAC,
                "discriminator_samples": int(ill.sum()), "last_frame": last, "beyond_the_taps": int(ill[:, F:].sum()),
                "max_abs_discriminator_diff": float(np.abs(D).max()) if D.size else 0.0,
                "pcm_samples_reached": int(reached.sum()), "pcm_samples_strict": int((~reached).sum()),
                "pcm_samples_reached_over_1_lsb": int(over.sum()), "max_abs_pcm_diff_lsb_reached": int(np.abs(d[reached]).max()) if reached.any() else 0,
                "max_abs_unexplained_lsb": float(np.abs((d - E)[reached & ~sat]).max()) if (reached & ~sat).any() else 0.0,
                "saturated_skipped": int((reached & sat).sum()),
                "samples_over_1_lsb": int(over.sum()),
                "why": "checked from a reset: polyphase windows still hold pre-stream zeros, arg() of a near-zero channel output is "
                       "ill-conditioned; the reference never restarts a stream and demodulates only squelch-opened channels "
                       "(src/sdr_pmr446.c:788, :834-836, :876-881)"}}


def fixtures(root):
    """(hp taps, de-emphasis b0, b1, a1) from the committed reference-derived fixtures (tests/golden: the reference's tap tables
    src/sdr_pmr446.c:56-136 and scripts/filter_des.py standard_deemph()), for audio_response()."""
    import os
    t = np.load(os.path.join(root, "tests", "golden", "pmr446_taps.npz"))
    de = np.load(os.path.join(root, "tests", "golden", "deemph_ref.npz"))
    b, a = de["b"].astype(np.float64), de["a"].astype(np.float64)
    return t["hp_audio_taps"].astype(np.float64), b[0] / a[0], b[1] / a[0], a[1] / a[0]


def option_taps(root):
    """(de-emphasis FIR taps, low-pass taps) of the reference's optional audio stages (src/sdr_pmr446.c:99-136), same fixture."""
    import os
    t = np.load(os.path.join(root, "tests", "golden", "pmr446_taps.npz"))
    return t["deemph_taps"].astype(np.float64), t["lp_audio_taps"].astype(np.float64)
