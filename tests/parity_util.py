"""Shared helpers for the parity tests: run oracle and HIP chain on the same blocks and compare."""
import numpy as np

from sdr_pmr446_amd import synth

# BASELINE.json configs (fs_in, M)
CFG_REF = (1.024e6, 16)     # the reference's own operating point (include/sdr_pmr446.h:13)
CFG2 = (2.4e6, 16)          # configs[0]/[1]
CFG3 = (61.44e6, 256)       # configs[2]/[3]
CFG5 = (1.0e9, 1024)        # configs[4]


def active_channels(M, synthesized=None, fs=None):
    """Channels compared against the oracle (synth.signal_channels): not empty, and -- when fs is given -- not inside the
    chain's own dc-block notch (at 1 GS/s the 80 kHz notch attenuates the ~8 channels around band centre by > 6 dB)."""
    return synth.signal_channels(M, fs, synthesized)


def run_blocks(chain, x, splits, want):
    """Feed x in consecutive blocks of the given sizes; concatenate per-output along time."""
    outs = {}
    pos = 0
    for n in splits:
        o = chain.process_block(x[pos:pos + n], want=want)
        pos += n
        for k, v in o.items():
            outs.setdefault(k, []).append(v)
    assert pos == len(x)
    res = {"n_frames": int(np.sum(outs["n_frames"]))}
    for k, v in outs.items():
        if k == "n_frames":
            continue
        if k == "rssi":
            res[k] = v
        elif k == "resampled":
            res[k] = np.concatenate(v)
        else:
            res[k] = np.concatenate(v, axis=1)
    return res


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def pcm_diff(a, b):
    return np.abs(a.astype(np.int32) - b.astype(np.int32))
