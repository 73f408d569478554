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
