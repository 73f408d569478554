"""CPU tier for SURVEY s8 row f3, the `dsd_in` chain (reference src/dsd_in.c:95-178): the oracle restatement against
known answers and a closed-form numpy model of the msresamp_rrrf interpolator, the product's host-side design and block
planner against the oracle, and the committed golden vector.  No GPU."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle
from sdr_pmr446_amd import chain, synth

FS = 1024000.0
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dsd_ref_point.npz")


def up_model(fm, o):
    """Closed form of msresamp_rrrf_execute (arbitrary resampler, then half-band interpolators) in float32,
    accumulating oldest sample first like the scalar liquid dot product."""
    step, S = o.info(2), o.info(1)
    bank = o.design(0).reshape(256, 14)
    n = len(fm)
    nu = ((n << 24) + step - 1) // step
    x = np.concatenate([np.zeros(13, np.float32), fm.astype(np.float32)])
    T = np.arange(nu, dtype=np.uint64) * np.uint64(step)
    q = (T >> np.uint64(24)).astype(np.int64)
    idx = ((T >> np.uint64(16)) & np.uint64(255)).astype(np.int64)
    u = np.zeros(nu, np.float32)
    for k in range(14):
        u = (u + bank[idx, k] * x[q + k]).astype(np.float32)     # x[q + k] = fm[q - 13 + k]
    for g in range(S):
        h1 = o.design(1 + g)
        m = len(h1) // 2
        xp = np.concatenate([np.zeros(2 * m, np.float32), u])
        i = np.arange(len(u))
        y0 = xp[i + 2 * m - m]
        y1 = np.zeros(len(u), np.float32)
        for k in range(2 * m):
            y1 = (y1 + h1[k] * xp[i + 1 + k]).astype(np.float32)  # u[i - 2m + 1 + k]
        u = np.stack([y0, y1], axis=1).reshape(-1)
    return u


def test_sizes_and_structure_of_the_reference_operating_point():
    o = oracle.OracleDsd()
    assert o.max_resampled == 4884 and o.max_out == 37511      # src/dsd_in.c:140-141 with SDR_INPUT_CHUNK 200000
    assert o.info(0) == 6                                        # 1.024 MS/s -> 12.5 kS/s: 6 half-band stages, r_a 0.78125
    assert o.info(1) == 1 and o.info(4) == 10                    # 3.84 = 1.92 x one half-band interpolator (m = 10)
    assert abs(o.info(2) - (1 << 24) / 1.92) <= 1.0


def test_interpolator_matches_closed_form_model():
    o = oracle.OracleDsd(max_block=60000)
    x = synth.synth_iq(150000, FS, 1, dev_hz=2500.0)
    fm, audio = [], []
    for a, b in ((0, 60000), (60000, 60001), (60001, 110000), (110000, 150000)):
        r = o.process_block(x[a:b], want=("fm", "audio"))
        fm.append(r["fm"]); audio.append(r["audio"])
    fm, audio = np.concatenate(fm), np.concatenate(audio)
    model = up_model(fm, o)
    assert len(model) == len(audio)
    assert np.array_equal(model, audio)


def test_fm_tone_known_answer():
    """A 1 kHz tone at 2.5 kHz deviation comes out as a 1 kHz sine at 48 kS/s, amplitude ~ 2 f_dev / 12.5 kHz."""
    o = oracle.OracleDsd()
    n = 600000
    t = np.arange(n) / FS
    iq = (0.5 * np.exp(1j * 2.5 * np.sin(2 * np.pi * 1000.0 * t))).astype(np.complex64)
    a = np.concatenate([o.process_block(iq[i:i + 200000], want=("audio",))["audio"] for i in range(0, n, 200000)])
    assert abs(len(a) / (n * 48000.0 / FS) - 1.0) < 2e-3
    seg = a[4000:4000 + 24000]
    spec = np.abs(np.fft.rfft(seg * np.hanning(len(seg))))
    assert abs(np.argmax(spec) * 48000.0 / len(seg) - 1000.0) <= 2.0
    amp = 2.0 * spec.max() / np.hanning(len(seg)).sum()
    assert 0.36 < amp < 0.41                                     # 0.4 less the sideband loss of the 12.5 kS/s stage


def test_carrier_offset_known_answer():
    """A carrier f Hz off centre demodulates to the constant 2 f / 12.5 kHz (freqdem kf = 0.5)."""
    o = oracle.OracleDsd()
    n = 400000
    iq = (0.4 * np.exp(2j * np.pi * 1500.0 * np.arange(n) / FS)).astype(np.complex64)
    a = np.concatenate([o.process_block(iq[i:i + 200000], want=("audio",))["audio"] for i in range(0, n, 200000)])
    assert np.abs(a[3000:] - 2 * 1500.0 / 12500.0).max() < 2e-3
    pcm = oracle.OracleDsd().process_block(iq[:200000])["pcm"]
    assert abs(int(pcm[-1]) - int(0.24 * 32767)) <= 70           # (int16_t)(x * INT16_MAX), src/dsd_in.c:174


def test_host_design_is_bit_identical_to_oracle():
    L = chain.load()
    for fs, au in ((FS, 48000.0), (2.4e6, 48000.0), (FS, 96000.0), (FS, 12500.0)):
        cfg = chain.make_dsd_cfg(fs_in=fs, audio_rate=au)
        o = oracle.OracleDsd(fs_in=fs, audio_rate=au)
        S = o.info(1)
        for what in [0, 1, 2, 3] + [4 + g for g in range(S)]:
            assert L.pmr_dsd_cfg_info(C.byref(cfg), what) == o.info(what), (fs, au, what)


def test_block_planner_matches_oracle_counts():
    L = chain.load()
    cfg = chain.make_dsd_cfg(max_block=70000)
    st = chain.DsdPlanState(0, 0, 0)
    o = oracle.OracleDsd(max_block=70000)
    rng = np.random.default_rng(5)
    for i in range(50):
        n = int(rng.integers(0, 70000)) if i % 6 else int(rng.integers(0, 70))
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.1
        r = o.process_block(x)
        ny, nz = C.c_uint(0), C.c_uint(0)
        assert L.pmr_dsd_plan_block(C.byref(cfg), C.byref(st), n, C.byref(ny), C.byref(nz)) == 0
        assert (ny.value, nz.value) == (r["n_resampled"], r["n_out"]), (i, n)


def test_create_fails_loudly_without_gpu_and_rejects_bad_cfg():
    import torch
    L = chain.load()
    bad = chain.make_dsd_cfg(audio_rate=8000.0)                  # decimating the audio is not a dsd_in configuration
    assert not L.pmr_dsd_create(C.byref(bad))
    if not torch.cuda.is_available():
        assert not L.pmr_dsd_create(C.byref(chain.make_dsd_cfg()))
        with pytest.raises(chain.PmrError):
            chain.PmrDsd()
    one = chain.make_cfg(FS, 1, 1000)                            # the one-channel front end is internal, not C-ABI
    assert not L.pmr_chain_create(C.byref(one))


def load_golden():
    g = np.load(GOLD)
    x = synth.synth_iq(int(g["n"]), float(g["fs"]), 1, dev_hz=float(g["dev_hz"]))
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["input_sha256"]), "synthetic generator drifted"
    return g, x


def test_oracle_reproduces_golden():
    g, x = load_golden()
    o = oracle.OracleDsd(fs_in=float(g["fs"]), max_block=int(max(g["splits"])))
    pcm, pos = [], 0
    for n in g["splits"]:
        pcm.append(o.process_block(x[pos:pos + int(n)])["pcm"]); pos += int(n)
    assert np.array_equal(np.concatenate(pcm), g["pcm"])
