"""include/pmr_mem.h: device memory + the synthetic SURVEY s8(d) stream generated in HBM by the library itself (bench.py's
input; stands in for the SoapySDR ingest, reference src/shared.c:62)."""
import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, pcm_diff
from sdr_pmr446_amd import synth

pytestmark = pytest.mark.gpu


def test_any_subrange_is_reproducible_and_periodic_blocks_repeat():
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n = 1 << 16
    whole = chain.synth_iq_device(2 * n, fs, M).download(np.complex64, 2 * n)
    a = chain.synth_iq_device(n, fs, M, n0=0).download(np.complex64, n)
    b = chain.synth_iq_device(n, fs, M, n0=n).download(np.complex64, n)
    assert np.array_equal(whole[:n], a) and np.array_equal(whole[n:], b)
    assert not np.array_equal(a, chain.synth_iq_device(n, fs, M, stream_id=1).download(np.complex64, n))
    # period_log2: the noiseless part of a block repeats exactly
    p0 = chain.synth_iq_device(n, fs, M, period_log2=16, snr_db=300.0, n0=0).download(np.complex64, n)
    p1 = chain.synth_iq_device(n, fs, M, period_log2=16, snr_db=300.0, n0=n).download(np.complex64, n)
    assert np.abs(p0 - p1).max() < 2e-6 and np.abs(p0).max() > 0.1


@pytest.mark.parametrize("cfg", [CFG2, CFG3], ids=["cfg2", "cfg3"])
def test_channel_plan_and_parity_on_the_device_generated_stream(cfg):
    """Channel k of the generated stream lands on channelizer output k (fm / carrier channels strong, every 8th empty), the FM
    tone has the analytic discriminator amplitude, and the HIP chain matches the oracle on it within +-1 LSB."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    n = 600000 if M == 16 else 1 << 23
    step = 1 if M == 16 else 5
    xd = chain.synth_iq_device(n, fs, M, dev_hz=500.0, channel_step=step)
    x = xd.download(np.complex64, n)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    rg, ro = g.process_block(x, want=("pcm", "rssi", "audio")), o.process_block(x, want=("pcm", "rssi"))
    ks = list(range(0, M, step))
    act = synth.signal_channels(M, fs, ks)
    assert pcm_diff(rg["pcm"][act], ro["pcm"][act]).max() <= 1
    empty = [k for k in range(M) if k not in act]
    assert np.min(rg["rssi"][act]) > np.max(rg["rssi"][empty]) + 15.0          # dB
    a = rg["audio"][:, 1000:].astype(np.float64)
    tt = np.arange(1000, rg["n_frames"]) / 12500.0
    for k in act[:12]:
        if synth.channel_kind(k) == "fm":
            fa = synth.audio_tone_hz(k)
            amp = 2.0 * abs(np.mean(a[k] * np.exp(-2j * np.pi * fa * tt)))
            w = 2 * np.pi * fa / 12500.0
            b0, a1 = 0.507301437230636, 0.014602874461272194
            de = abs(b0 * (1 + np.exp(-1j * w)) / (1 + a1 * np.exp(-1j * w)))
            expect = 4.0 * (2 * 500.0 / 12500.0) * de * np.sin(w / 2) / (w / 2)     # first difference of the phase: sinc droop
            assert abs(amp - expect) < 0.04 * expect, (k, amp, expect)
    g.close(); o.close(); xd.free()
