"""The conditioning rule of sdr_pmr446_amd/parity_rule.py (how bench.py and tests/test_gpu_streams.py hold PCM against the oracle when the
comparison starts at a reset), on synthetic arrays and on a real oracle run: only the start-up is ill-conditioned, a PCM deviation is
accepted only where the measured discriminator difference AT ILL-CONDITIONED samples explains it through the audio filter, and the
filter response used for that is the oracle's own audio path."""
import os

import numpy as np

import oracle
from sdr_pmr446_amd import parity_rule as pr, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _h():
    hp, b0, b1, a1 = pr.fixtures(ROOT)
    return pr.audio_response(hp, 4.0, b0, b1, a1)


def _chan(K=3, T=1200, ramp=8):
    r = np.ones((K, T), np.complex64)
    r[:, :ramp] = 1e-4                        # the bank's windows still hold pre-stream zeros
    return r


def test_ill_conditioned_class():
    ill = pr.ill_conditioned(_chan())
    assert ill[:, :9].all() and not ill[:, 9:].any()           # frame 8 is ill through its r' = frame 7


def test_branch_cut_class():
    """A phase advance of exactly pi per frame (energy at half the channel rate from the channel's centre: what the dc blocker's
    start-up transient does to the two channels next to band centre): +pi and -pi are the same angle, the discriminator outputs 2 apart."""
    c = np.ones((2, 600), np.complex64)
    c[0, 40:60] = (-1.0) ** np.arange(20)                       # r alternates sign: conj(r') r = -1 exactly
    ill = pr.ill_conditioned(c)
    assert ill[0, 41:61].all() and not ill[0, 61:].any() and not ill[0, 1:41].any() and not ill[1, 1:].any()
    h = _h()
    ref = np.zeros(c.shape, np.int32)
    fm_ref = np.zeros((2, 100)); fm_got = fm_ref.copy()
    fm_ref[0, 45], fm_got[0, 45] = 1.0, -1.0                    # the oracle says +pi, the chain -pi
    E = np.zeros(600); E[45:45 + len(h)] = 32767 * -2.0 * h
    got = ref.copy(); got[0] = np.trunc(E).astype(np.int32)
    got = np.clip(got, -32767, 32767)
    v = pr.check(got, ref, c, fm_got, fm_ref, h)
    assert v["ok"], v
    assert not pr.check(got, ref, np.ones_like(c), fm_got, fm_ref, h)["ok"]      # the same PCM with no sample on the cut: unexplained


def test_verdicts():
    c, h = _chan(), _h()
    K, T = c.shape
    F = 40
    ref = np.zeros((K, T), np.int32)
    fm_ref = np.zeros((K, F))
    # a discriminator difference of 0.3 at the ill frame 3 of channel 0: the PCM must differ by the filter's response to it
    fm_got = fm_ref.copy(); fm_got[0, 3] = 0.3
    E = np.zeros(T); E[3:3 + len(h)] = 32767 * 0.3 * h
    got = ref.copy(); got[0] = np.trunc(E).astype(np.int32)
    v = pr.check(got, ref, c, fm_got, fm_ref, h)
    assert v["ok"] and v["max_abs_pcm_diff_lsb"] == 0 and v["ill_conditioned"]["max_abs_pcm_diff_lsb_reached"] > 10000, v
    assert v["ill_conditioned"]["max_abs_unexplained_lsb"] < 1.0
    # the same PCM WITHOUT the discriminator difference to explain it fails
    assert not pr.check(got, ref, c, fm_ref, fm_ref, h)["ok"]
    # ... and so does a discriminator difference at a WELL-conditioned frame (it explains nothing)
    fm_w = fm_ref.copy(); fm_w[0, 20] = 0.3
    Ew = np.zeros(T); Ew[20:20 + len(h)] = 32767 * 0.3 * h
    gw = ref.copy(); gw[0] = np.trunc(Ew).astype(np.int32)
    assert not pr.check(gw, ref, c, fm_w, fm_ref, h)["ok"]
    # 3 LSB unexplained inside the reached span fails; 1 LSB passes; steady state: the one-line bar
    g2 = got.copy(); g2[0, 100] += 3
    assert not pr.check(g2, ref, c, fm_got, fm_ref, h)["ok"]
    g2 = got.copy(); g2[0, 100] += 1; g2[1, 50] = 1; g2[2, 900] = -1
    assert pr.check(g2, ref, c, fm_got, fm_ref, h)["ok"]
    g2 = got.copy(); g2[2, 700] = 2
    v = pr.check(g2, ref, c, fm_got, fm_ref, h)
    assert not v["ok"] and v["max_abs_pcm_diff_lsb"] == 2
    # malformed inputs are failures, not exceptions
    assert not pr.check(got[:, :-1], ref, c, fm_got, fm_ref, h)["ok"]
    assert not pr.check(got, ref, c, fm_got[:, :2], fm_ref[:, :2], h)["ok"]          # taps that miss the ill frame 3: nothing explains the PCM


def test_audio_response_is_the_oracles_audio_path_and_only_the_start_up_is_ill_conditioned():
    fs, M, n = 2.4e6, 16, 1 << 20
    x = synth.synth_iq(n, fs, M, stream_id=3)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    r = o.process_block(x, want=("pcm", "chan", "fm", "audio"))
    o.close()
    act = synth.signal_channels(M, fs)
    ill = pr.ill_conditioned(r["chan"][act])
    assert ill.any() and np.nonzero(ill.any(axis=0))[0].max() < pr.PFB_FRAMES
    # linearity: the oracle's audio is its discriminator output through audio_response() (f32 vs f64: 1e-5 of the scale)
    h = _h()
    k = act[0]
    pred = np.convolve(r["fm"][k].astype(np.float64), h)[:r["fm"].shape[1]]
    assert np.abs(pred - r["audio"][k]).max() <= 1e-5 * max(1.0, np.abs(r["audio"][k]).max())
    assert 1.5 < np.abs(h).max() < 2.2 and np.argmax(np.abs(h)) in (188, 189)


def test_audio_response_with_the_optional_stages_is_the_oracles_audio_path():
    """lowpass (:453-454, :900-902) and the FIR form of the de-emphasis (:457-458): the responses tools/soak.py hands the rule."""
    fs, M, n = 2.4e6, 16, 1 << 19
    x = synth.synth_iq(n, fs, M, stream_id=1)
    hp, b0, b1, a1 = pr.fixtures(ROOT)
    de_fir, lp = pr.option_taps(ROOT)
    for opts in (dict(lowpass=True), dict(deemph_fir=True), dict(lowpass=True, deemph_fir=True)):
        o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n, **opts)
        r = o.process_block(x, want=("fm", "audio"))
        o.close()
        h = pr.audio_response(hp, 4.0, b0, b1, a1, deemph_fir_taps=de_fir if opts.get("deemph_fir") else None, lp_taps=lp if opts.get("lowpass") else None)
        k = synth.signal_channels(M, fs)[0]
        pred = np.convolve(r["fm"][k].astype(np.float64), h)[:r["fm"].shape[1]]
        assert np.abs(pred - r["audio"][k]).max() <= 2e-5 * max(1.0, np.abs(r["audio"][k]).max()), opts
