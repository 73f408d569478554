#!/usr/bin/env python3
"""Generate tests/golden/chain_*.npz: small golden vectors for the hot path.

The reference ships no fixtures and cannot run here (liquid-dsp v1.7.0 is absent), so these vectors are produced
by the CPU oracle (oracle/) on the deterministic synthetic IQ of sdr_pmr446_amd/synth.py.  They pin the oracle
against accidental change (CPU tier, bit-exact) and give the HIP path a committed target (GPU tier, +-1 LSB).
Inputs are regenerated from the recorded synth parameters; their sha256 is stored to detect generator drift.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sdr_pmr446_amd import synth  # noqa: E402

CASES = {
    # name: dict(fs, M, n, splits, dev_hz, opts)
    "cfg2_default": dict(fs=2.4e6, M=16, n=100000, splits=[100000], dev_hz=500.0, opts={}),
    "cfg2_split_lp": dict(fs=2.4e6, M=16, n=120000, splits=[50000, 1, 30000, 39999], dev_hz=2500.0,
                          opts=dict(lowpass=True)),
    "ref_point": dict(fs=1.024e6, M=16, n=100000, splits=[100000], dev_hz=500.0, opts={}),
}


def run_case(c):
    x = synth.synth_iq(c["n"], c["fs"], c["M"], dev_hz=c["dev_hz"])
    ch = oracle.OracleChain(fs_in=c["fs"], num_channels=c["M"], max_block=max(c["splits"]), **c["opts"])
    pcm, chan, pos = [], [], 0
    for n in c["splits"]:
        o = ch.process_block(x[pos:pos + n], want=("pcm", "chan"))
        pcm.append(o["pcm"]); chan.append(o["chan"]); pos += n
    return x, np.concatenate(pcm, axis=1), np.concatenate(chan, axis=1)


def main():
    for name, c in CASES.items():
        x, pcm, chan = run_case(c)
        path = os.path.join(ROOT, "tests", "golden", "chain_%s.npz" % name)
        np.savez_compressed(path, pcm=pcm, chan_head=chan[:, :64], input_sha256=hashlib.sha256(x.tobytes()).hexdigest(),
                            fs=c["fs"], M=c["M"], n=c["n"], splits=np.array(c["splits"]), dev_hz=c["dev_hz"],
                            lowpass=int(c["opts"].get("lowpass", False)))
        print(path, pcm.shape, os.path.getsize(path))


def main_dsd():
    """dsd_in chain: one FM channel at band centre, reference operating point, ragged block split."""
    fs, n, splits, dev = 1.024e6, 260000, [200000, 1, 59999], 2500.0
    x = synth.synth_iq(n, fs, 1, dev_hz=dev)
    o = oracle.OracleDsd(fs_in=fs, max_block=max(splits))
    pcm, pos = [], 0
    for k in splits:
        pcm.append(o.process_block(x[pos:pos + k])["pcm"]); pos += k
    pcm = np.concatenate(pcm)
    path = os.path.join(ROOT, "tests", "golden", "dsd_ref_point.npz")
    np.savez_compressed(path, pcm=pcm, input_sha256=hashlib.sha256(x.tobytes()).hexdigest(), fs=fs, n=n,
                        splits=np.array(splits), dev_hz=dev)
    print(path, pcm.shape, os.path.getsize(path))


if __name__ == "__main__":
    main()
    main_dsd()
