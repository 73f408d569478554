"""Vectors produced by the REFERENCE'S OWN CODE (tools/make_ref_fixtures.py, run in the build container where /root/reference
exists; only the resulting data is committed):

  tests/golden/ctcss_ref.npz   the CTCSS detector -- src/sdr_pmr446.c:338-409 (ctcss_detector_reset / _create / _analyze) and its
                               struct include/sdr_pmr446.h:42-52, cut out of the reference and compiled with gcc -- run on the
                               detector inputs of a tone-level sweep that straddles both decision thresholds (:403-404);
  tests/golden/deemph_ref.npz  standard_deemph() of scripts/filter_des.py:31-44, imported and evaluated;
  tests/golden/rssi_ref.npz    average_power() -- src/sdr_pmr446.c:330-336, cut out and compiled -- on channelizer output rows.

They pin the pieces of the oracle that are not restatements of liquid-dsp: orc_chain.c's detector (the thing the GPU detector is
compared with), the de-emphasis coefficients and the RSSI arithmetic (what the GPU's rssi_db is compared with)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EV = np.dtype([("index", np.int32), ("detected", np.int32), ("max_power", np.float32), ("avg_power", np.float32)])


def _detector(x, cap=16):
    """oracle/orc_chain.c's detector on one float32 stream from zero state: events[B], powers[B][38]"""
    L = oracle.lib()
    L.orc_ctcss_detector_run.argtypes = [C.c_void_p, C.c_uint, C.c_double, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p]
    L.orc_ctcss_detector_run.restype = C.c_uint
    x = np.ascontiguousarray(x, dtype=np.float32)
    ev = np.zeros(cap, dtype=EV)
    pw = np.zeros((cap, 38), dtype=np.float32)
    n = L.orc_ctcss_detector_run(x.ctypes.data, len(x), 12500.0, 2441, ev.ctypes.data, cap, pw.ctypes.data)
    return ev[:n], pw[:n]


def _dcblock(x):
    L = oracle.lib()
    L.orc_dcblock_rrrf_run.argtypes = [C.c_void_p, C.c_uint, C.c_float, C.c_void_p]
    L.orc_dcblock_rrrf_run.restype = None
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    L.orc_dcblock_rrrf_run(x.ctypes.data, len(x), 0.0005, y.ctypes.data)
    return y


def test_oracle_ctcss_detector_equals_the_references_code_on_the_same_samples():
    g = np.load(os.path.join(GOLD, "ctcss_ref.npz"))
    B = g["index"].shape[1]
    assert B >= 4 and g["x"].shape[1] == B * 2441
    for row, k in enumerate(g["x_channels"]):
        ev, pw = _detector(g["x"][row])
        assert len(ev) == B
        assert np.array_equal(ev["index"], g["index"][k]) and np.array_equal(ev["detected"], g["detected"][k])
        # same float32 recurrence, same order of operations: the Goertzel powers agree to the last bit
        assert np.array_equal(pw, g["power"][k]) and np.array_equal(ev["max_power"], g["max_power"][k])
    det = g["detected"][:, 1:]
    assert det.sum() >= 8 and (1 - det).sum() >= 8             # the sweep exercises both outcomes of :403-404


def test_oracle_chain_reaches_the_references_decisions_on_the_synthetic_sweep():
    """End to end on the CPU side: synthetic IQ -> oracle chain's low-pass branch (:884-889) -> dc blocker (:606) -> detector must
    arrive at the decisions the reference's detector code took on that stream (all 16 channels), and the chain's own `ctcss`
    events (the comparison target of tests/test_gpu_ctcss.py) are those decisions."""
    from sdr_pmr446_amd import synth
    g = np.load(os.path.join(GOLD, "ctcss_ref.npz"))
    fs, M, n = float(g["synth_fs"]), int(g["synth_M"]), int(g["synth_n"])
    devs = list(g["synth_ctcss_devs"])
    x = synth.synth_iq(n, fs, M, dev_hz=float(g["synth_dev_hz"]), ctcss_dev_of=lambda k: devs[k])
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    r = o.process_block(x, want=("pcm", "ctcss_lp", "ctcss"))
    o.close()
    B = g["index"].shape[1]
    assert r["ctcss"].shape == (M, B)
    for k in range(M):
        ev, _ = _detector(_dcblock(r["ctcss_lp"][k]))
        for src in (ev, r["ctcss"][k]):
            assert np.array_equal(src["index"], g["index"][k]), k
            assert np.array_equal(src["detected"], g["detected"][k]), k
            assert np.array_equal(src["max_power"], g["max_power"][k]), k
    for row, k in enumerate(g["x_channels"]):                 # and the stored detector inputs are what the chain produces
        assert np.array_equal(_dcblock(r["ctcss_lp"][k])[:B * 2441], g["x"][row])


def test_deemphasis_coefficients_are_the_design_scripts():
    """reference scripts/filter_des.py:31-44 standard_deemph() -> the literals at src/sdr_pmr446.c:462-463 -> the oracle's and the
    library's design (row a8)."""
    g = np.load(os.path.join(GOLD, "deemph_ref.npz"))
    b, a = g["b"], g["a"]
    assert float(g["tau"]) == 50e-6 and float(g["fs"]) == 12500.0
    assert b[0] == b[1] == 0.507301437230636 and a[0] == 1.0 and a[1] == 0.014602874461272194      # :462-463, digit for digit
    L = oracle.lib()
    L.orc_deemph_iir_coefs.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_deemph_iir_coefs.restype = None
    ob, oa = np.zeros(2, np.float32), np.zeros(2, np.float32)
    L.orc_deemph_iir_coefs(ob.ctypes.data, oa.ctypes.data)
    assert np.array_equal(ob, b.astype(np.float32)) and np.array_equal(oa, a.astype(np.float32))
    # the library's own design (host-only helper: no GPU needed), normalised by a0 like liquid's iirfilt
    from sdr_pmr446_amd import chain
    cfg = chain.make_cfg()
    n = chain.load().pmr_cfg_design(C.byref(cfg), chain.DESIGN_DEEMPH, 0, None, 0)
    de = np.zeros(n, dtype=np.float32)
    chain.load().pmr_cfg_design(C.byref(cfg), chain.DESIGN_DEEMPH, 0, de.ctypes.data, n)
    assert n == 3 and np.array_equal(de, np.array([b[0] / a[0], b[1] / a[0], a[1] / a[0]], dtype=np.float32))


def test_oracle_rssi_equals_the_references_average_power_bit_for_bit():
    """tests/golden/rssi_ref.npz: average_power() (src/sdr_pmr446.c:330-336, cut out of the reference and compiled) on the channelizer
    output rows of a synthetic block -- the numbers find_max_rssi_channel (:668-700) compares.  The oracle chain, run on the same IQ,
    must return the same tap-off and the same 16 dB values bit for bit; the reference's max - mean selection on them is what
    oracle/squelch.py and the product's pmr_find_max_rssi_channel compute."""
    from oracle import squelch as orc_sq
    from sdr_pmr446_amd import chain, synth
    g = np.load(os.path.join(GOLD, "rssi_ref.npz"))
    fs, M, n = float(g["synth_fs"]), int(g["synth_M"]), int(g["synth_n"])
    x = synth.synth_iq(n, fs, M, dev_hz=float(g["synth_dev_hz"]))
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    r = o.process_block(x, want=("pcm", "chan", "rssi"))
    o.close()
    assert np.array_equal(r["chan"], g["chan"])
    assert np.array_equal(np.asarray(r["rssi"], dtype=np.float32), g["rssi_db"])
    ref = g["rssi_db"]
    # :668-700 on the reference's numbers: max channel and (max - mean of the dB values), all channels enabled
    want_i = int(np.argmax(ref))
    want_v = np.float32(ref[want_i] - np.float32(np.float32(ref.sum(dtype=np.float32)) / np.float32(M)))
    oi, ov = orc_sq.find_max_rssi_channel(ref, (1 << M) - 1)
    assert oi == want_i and abs(float(ov) - float(want_v)) < 1e-4
    L = chain.load(build_if_missing=True)
    mr = C.c_float(0.0)
    assert L.pmr_find_max_rssi_channel(np.ascontiguousarray(ref).ctypes.data, M, None, 0, C.byref(mr)) == want_i
    assert abs(mr.value - float(ov)) < 1e-5
