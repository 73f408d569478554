"""Committed golden vectors (tests/golden/chain_*.npz, made by tools/make_golden.py from the CPU oracle):
CPU tier -- the oracle reproduces them bit for bit; GPU tier -- the HIP path is within +-1 LSB of them."""
import glob
import hashlib
import os

import numpy as np
import pytest

import oracle
from parity_util import active_channels, pcm_diff
from sdr_pmr446_amd import synth

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_*.npz")))


def _load(path):
    g = np.load(path)
    x = synth.synth_iq(int(g["n"]), float(g["fs"]), int(g["M"]), dev_hz=float(g["dev_hz"]))
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["input_sha256"]), "synthetic generator drifted"
    return g, x


def _run(ch, x, splits):
    pcm, chan, pos = [], [], 0
    for n in splits:
        o = ch.process_block(x[pos:pos + int(n)], want=("pcm", "chan"))
        pcm.append(o["pcm"]); chan.append(o["chan"]); pos += int(n)
    return np.concatenate(pcm, axis=1), np.concatenate(chan, axis=1)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_golden(path):
    g, x = _load(path)
    ch = oracle.OracleChain(fs_in=float(g["fs"]), num_channels=int(g["M"]), max_block=int(max(g["splits"])),
                            lowpass=bool(g["lowpass"]))
    pcm, chan = _run(ch, x, g["splits"])
    assert np.array_equal(pcm, g["pcm"]) and np.array_equal(chan[:, :64], g["chan_head"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_hip_matches_golden(path):
    from sdr_pmr446_amd import chain
    g, x = _load(path)
    M = int(g["M"])
    ch = chain.PmrChain(fs_in=float(g["fs"]), num_channels=M, max_block=int(max(g["splits"])),
                        lowpass=bool(g["lowpass"]))
    pcm, chan = _run(ch, x, g["splits"])
    act = active_channels(M)
    assert pcm.shape == g["pcm"].shape
    assert pcm_diff(pcm[act], g["pcm"][act]).max() <= 1          # north_star: int16 PCM within +-1 LSB
    ref = g["chan_head"]
    assert np.abs(chan[:, :64] - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_hip_rssi_matches_the_references_average_power():
    """the GPU's per-channel RSSI (reduced inside the channelizer, finished in the FIR launch) against the numbers
    the REFERENCE's own average_power() (src/sdr_pmr446.c:330-336, cut out and compiled: tests/golden/rssi_ref.npz) returns for the
    same block, and the selection find_max_rssi_channel (:668-700) makes on them."""
    import ctypes as C
    from sdr_pmr446_amd import chain
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rssi_ref.npz"))
    fs, M, n = float(g["synth_fs"]), int(g["synth_M"]), int(g["synth_n"])
    x = synth.synth_iq(n, fs, M, dev_hz=float(g["synth_dev_hz"]))
    ch = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    r = ch.process_block(x, want=("pcm", "chan", "rssi"))
    ref = g["rssi_db"]
    assert r["chan"].shape == g["chan"].shape
    assert np.abs(r["chan"] - g["chan"]).max() <= 2e-5 * np.abs(g["chan"]).max()
    assert np.abs(np.asarray(r["rssi"]) - ref).max() <= 2e-3               # dB; float32 sums of 520 magnitudes in another order
    mr = C.c_float(0.0)
    rs = np.ascontiguousarray(r["rssi"], dtype=np.float32)
    assert ch._L.pmr_find_max_rssi_channel(rs.ctypes.data, M, None, 0, C.byref(mr)) == int(np.argmax(ref))
    ch.close()
