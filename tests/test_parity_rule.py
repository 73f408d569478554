"""The conditioning rule of sdr_pmr446_amd/parity_rule.py (how bench.py and tests/test_gpu_streams.py hold PCM against the oracle when the
comparison starts at a reset), on synthetic arrays and on a real oracle run: the relaxed classes are the start-up only, they are placed
where the audio FIR carries an ill-conditioned discriminator sample, and a deviation outside them fails at 2 LSB."""
import numpy as np

import oracle
from sdr_pmr446_amd import parity_rule as pr, synth


def _chan(K=3, T=1200, ramp=8):
    r = np.ones((K, T), np.complex64)
    r[:, :ramp] = 1e-4                        # the bank's windows still hold pre-stream zeros
    return r


def test_classes_sit_where_the_fir_carries_the_ill_samples():
    c = _chan()
    centre, span, ill = pr.classify(c)
    assert ill[:, :9].all() and not ill[:, 9:].any()           # frame 8 is ill through its r' = frame 7
    lo, hi = pr.CENTRE_LAGS
    assert centre[:, lo:8 + hi + 1].all() and not centre[:, :lo].any() and not centre[:, 8 + hi + 1:].any()
    assert span[:, :lo].all() and span[:, 8 + hi + 1:8 + pr.FIR_TAPS].all() and not span[:, 8 + pr.FIR_TAPS:].any()
    assert not (centre & span).any()


def test_verdicts():
    c = _chan()
    ref = np.zeros(c.shape, np.int32)
    got = ref.copy()
    assert pr.check(got, ref, c)["ok"]
    got[0, 198] = 3                                            # round 5's case: ill frame 10-ish + lag 190
    v = pr.check(got, ref, c)
    assert v["ok"] and v["ill_conditioned"]["samples_over_1_lsb_centre"] == 1 and v["max_abs_pcm_diff_lsb"] == 0
    got[0, 198] = 5
    assert not pr.check(got, ref, c)["ok"]
    got[0, 198] = 0; got[1, 50] = 2                            # off-centre lag: 2 allowed, 3 not
    assert pr.check(got, ref, c)["ok"]
    got[1, 50] = 3
    assert not pr.check(got, ref, c)["ok"]
    got[1, 50] = 0; got[2, 700] = 2                            # steady state: the one-line bar
    v = pr.check(got, ref, c)
    assert not v["ok"] and v["max_abs_pcm_diff_lsb"] == 2
    assert not pr.check(got[:, :-1], ref, c)["ok"]             # shape mismatch is a failure, not an exception


def test_on_a_real_stream_only_the_start_up_is_ill_conditioned():
    fs, M, n = 2.4e6, 16, 1 << 20
    x = synth.synth_iq(n, fs, M, stream_id=3)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    r = o.process_block(x, want=("pcm", "chan"))
    o.close()
    act = synth.signal_channels(M, fs)
    centre, span, ill = pr.classify(r["chan"][act])
    assert ill.any() and np.nonzero(ill.any(axis=0))[0].max() < pr.PFB_FRAMES
    T = r["chan"].shape[1]
    assert T > 2000 and not (centre | span)[:, pr.PFB_FRAMES + pr.FIR_TAPS:].any()
    # far fewer relaxed samples than round 5's blanket window (409 frames of every channel at <= 8 LSB): the 4-LSB class is a few dozen frames
    assert centre.sum() < 45 * len(act)
