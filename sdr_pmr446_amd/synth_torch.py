"""Synthetic IQ generated directly in HBM with torch (bench input; same channel plan as synth.py).

Not bit-identical to synth.synth_iq (noise comes from torch's generator); parity tests always feed the SAME
array to both implementations, so that does not matter.  PyTorch is used only as a device-memory allocator and
random source here.
"""
import math

import torch

from .synth import CHANNEL_WIDTH_HZ, CTCSS_FREQS, SEED_BASE, audio_tone_hz, channel_kind


def synth_iq_torch(n, fs_in, num_channels, device, stream_id=0, snr_db=30.0, dev_hz=2500.0, ctcss_dev_hz=300.0,
                   chunk_elems=1 << 24, channels=None, periodic=False):
    """complex64 tensor [n] on `device` (viewable as float32 [n, 2], i.e. interleaved cf32).
    channels: channel indices to synthesise (default all M); amplitudes / noise do not depend on it.
    periodic: snap every carrier / tone frequency to the grid fs_in / n (< fs_in / 2n off, i.e. a few Hz), so that the block
    REPEATED back to back is one phase-continuous stream (bench.py feeds the same block every step; without this every block
    boundary is a phase jump in all carriers -- a click the discriminator sees)."""
    M = num_channels
    g = torch.Generator(device=device)
    g.manual_seed((SEED_BASE + stream_id) & 0x7FFFFFFFFFFFFFFF)
    amp = 0.5 / math.sqrt(M)
    sigma = math.sqrt(amp * amp / (10.0 ** (snr_db / 10.0)) * (fs_in / CHANNEL_WIDTH_HZ))
    ks = [k for k in (range(M) if channels is None else channels) if channel_kind(k) != "empty"]
    fk = torch.tensor([(k - (M - 1) / 2.0) * CHANNEL_WIDTH_HZ for k in ks], dtype=torch.float64, device=device)
    fm_on = torch.tensor([1.0 if channel_kind(k) == "fm" else 0.0 for k in ks], dtype=torch.float64, device=device)
    fa = torch.tensor([audio_tone_hz(k) for k in ks], dtype=torch.float64, device=device)
    fc = torch.tensor([float(CTCSS_FREQS[k % 38]) for k in ks], dtype=torch.float64, device=device)
    if periodic:
        grid = fs_in / n
        fk, fa, fc = (torch.round(v / grid) * grid for v in (fk, fa, fc))
    ph0 = torch.rand(len(ks), generator=g, device=device, dtype=torch.float64) * (2 * math.pi)
    out = torch.empty(n, dtype=torch.complex64, device=device)
    chunk = max(1024, chunk_elems // max(1, len(ks)))
    two_pi = 2.0 * math.pi
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        t = torch.arange(c0, c1, dtype=torch.float64, device=device) / fs_in          # [L]
        cyc = fk[:, None] * t[None, :]                                                  # carrier cycles
        cyc = cyc - torch.floor(cyc)
        ph = two_pi * cyc + ph0[:, None]
        ph = ph + fm_on[:, None] * ((dev_hz / fa)[:, None] * torch.sin(two_pi * fa[:, None] * t[None, :])
                                    + (ctcss_dev_hz / fc)[:, None] * torch.sin(two_pi * fc[:, None] * t[None, :]))
        ph32 = torch.remainder(ph, two_pi).to(torch.float32)
        re = torch.cos(ph32).sum(0) * amp
        im = torch.sin(ph32).sum(0) * amp
        nz = torch.randn(c1 - c0, 2, generator=g, device=device, dtype=torch.float32) * (sigma / math.sqrt(2.0))
        out[c0:c1] = torch.complex(re + nz[:, 0], im + nz[:, 1])
    return out
