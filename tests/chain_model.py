"""Closed-form (index-arithmetic) float64 model of every stage of the hot path.

The oracle (oracle/orc_*.c) restates the reference as liquid-style *sequential objects* (push / execute,
ring buffers, per-sample state).  The HIP kernels cannot run that way: they need each output as an explicit
sum over absolute input indices.  This module writes those sums down once, in numpy float64, from a stream
start with zero state.  tests/test_oracle_model.py checks the formulas against the sequential oracle; the
kernels in sdr_pmr446_amd/csrc implement exactly these formulas in float32.

Notation (all indices absolute from stream start / last reset):
  x[n]                       raw cf32 input                                   (src/sdr_pmr446.c:789)
  yb[n] = x[n] - alpha_eff * v[n-1],  v[n] = x[n] + lam * v[n-1]              (:795)
  z_{e+1}[i] = sum_k hb_e[k] * z_e[2i+1-k],  z_0 = yb,  e = execution order   (:796)
  dec[q] = z_h[q] / 2^h
  out[j] = sum_{n<2m} hA[idx_j + npfb*n] * dec[q_j - n],  q_j = (j*step)>>24, idx_j = ((j*step)&0xffffff)>>16
  xm[s]  = out[s] * exp(-i*2*pi*(s*dtheta mod 2^32)/2^32)                     (:808-812)
  X_c[t] = sum_{n<p} h[(M-1-c) + n*M] * xm[(t-n)*M + c]                       (:814)
  y[t,:] = FFT_forward(X[t,:])
  fm_k[t] = arg(conj(y_k[t-1]) * y_k[t]) / (2*pi*kf)                          (:881)
  hp_k[t] = sum_i hp[i] * fm_k[t-i]                                           (:882)
  a_k[t]  = deemph(gain * hp_k[t])  (IIR b0(u[t]+u[t-1]) - a1*a[t-1], or FIR) (:890-899)
  pcm     = sat(trunc(a * 32767))                                             (src/dsd_in.c:174)
"""
import numpy as np
from scipy.signal import lfilter


def dcblock(x, alpha=np.float32(0.0005)):
    a1 = np.float64(np.float32(-1.0) + np.float32(alpha))
    return lfilter([1.0, -1.0], [1.0, a1], x.astype(np.complex128))


def halfband_cascade(yb, hb, num_stages):
    """hb[g] = prototype of design stage g; stage num_stages-1 executes first."""
    z = yb
    for e in range(num_stages):
        h = hb[num_stages - 1 - e].astype(np.float64)
        n_out = len(z) // 2
        c = np.convolve(z, h)               # c[n] = sum_k h[k] z[n-k]
        z = c[1:2 * n_out:2]                # z1[i] = c[2i+1]
    return z / float(1 << num_stages)


def arb_resample(dec, proto, step, npfb=256, m=7):
    Q = len(dec)
    ny = -(-(Q << 24) // step)              # ceil(Q*2^24/step): outputs with q_j < Q
    j = np.arange(ny, dtype=np.int64)
    ph = j * np.int64(step)
    qj = ph >> 24
    idx = (ph & 0xFFFFFF) >> (24 - int(np.log2(npfb)))
    out = np.zeros(ny, dtype=np.complex128)
    pr = proto.astype(np.float64)
    for n in range(2 * m):
        src = qj - n
        ok = src >= 0
        out[ok] += pr[idx[ok] + npfb * n] * dec[src[ok]]
    return out


def nco_mix(xr, dtheta, s0=0):
    s = (np.arange(len(xr), dtype=np.uint64) + np.uint64(s0))
    th = (s * np.uint64(dtheta)) & np.uint64(0xFFFFFFFF)
    return xr * np.exp(-2j * np.pi * th.astype(np.float64) / 4294967296.0)


def channelize(xm, h, M, p):
    """Returns y[t, k] for complete frames."""
    T = len(xm) // M
    fr = xm[:T * M].reshape(T, M)
    hh = h.astype(np.float64)
    X = np.zeros((T, M), dtype=np.complex128)
    for n in range(p):
        taps = hh[(M - 1 - np.arange(M)) + n * M]        # per branch c
        if n == 0:
            X += fr * taps
        elif n < T:
            X[n:] += fr[:T - n] * taps
    return np.fft.fft(X, axis=1)


def freqdem(y, kf=0.5):
    prev = np.vstack([np.zeros((1, y.shape[1]), dtype=y.dtype), y[:-1]])
    return np.angle(np.conj(prev) * y) / (2 * np.pi * kf)


def audio_chain(fm, hp, gain=4.0, lowpass=False, lp=None, deemph_fir=False, deemph_taps=None):
    """fm: [T, M] -> float audio [T, M] (and the ctcss low-pass branch)."""
    hpf = lfilter(hp.astype(np.float64), [1.0], fm, axis=0)
    d = (len(hp) - 1) // 2
    delayed = np.vstack([np.zeros((d, fm.shape[1])), fm[:fm.shape[0] - d]]) if fm.shape[0] > d else np.zeros_like(fm)
    ctcss_lp = delayed - hpf
    u = hpf * gain
    if deemph_fir:
        a = lfilter(deemph_taps.astype(np.float64), [1.0], u, axis=0)
    else:
        b0 = np.float64(np.float32(0.507301437230636))
        a1 = np.float64(np.float32(0.014602874461272194))
        a = lfilter([b0, b0], [1.0, a1], u, axis=0)
    if lowpass:
        a = lfilter(lp.astype(np.float64), [1.0], a, axis=0)
    return a, ctcss_lp


def pcm_from_float(a):
    s = np.asarray(a, dtype=np.float32) * np.float32(32767.0)
    s = np.clip(s, -32768.0, 32767.0)
    return np.trunc(s).astype(np.int16)


def run_model(x, design, M, hp, alpha=np.float32(0.0005), gain=4.0, kf=0.5, **audio_kw):
    """Full chain from zero state.  design = OracleChain.design_dict().  Returns dict of stage outputs."""
    yb = dcblock(x, alpha)
    dec = halfband_cascade(yb, design["hb"], design["num_stages"])
    xr = arb_resample(dec, design["arb"], design["arb_step"], design["arb_npfb"], design["arb_m"])
    xm = nco_mix(xr, design["nco_dtheta"])
    y = channelize(xm, design["pfb"], M, design["pfb_p"])
    fm = freqdem(y, kf)
    audio, ctcss_lp = audio_chain(fm, hp, gain, **audio_kw)
    return {"resampled": xr, "chan": y.T, "fm": fm.T, "audio": audio.T, "ctcss_lp": ctcss_lp.T}
