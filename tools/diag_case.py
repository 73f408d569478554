#!/usr/bin/env python3
"""GPU box: one (fs, M, options, splits) case of tools/soak.py through chain and oracle with every tap compared per block -- where does a
failing case first part from the oracle?   python3 tools/diag_case.py fs M As max_block split,split,..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from sdr_pmr446_amd import chain, synth
fs, M, As, mb = float(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
splits = [int(t) for t in sys.argv[5].split(",")]
n = sum(splits)
ks = None if M <= 64 else list(range(0, M, M // 16))
x = synth.synth_iq(n, fs, M, channels=ks, dev_hz=1500.0, dc_offset=0.003)
g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb, resamp_As=As)
o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb, resamp_As=As)
print("plans fe/chan/fir", g.info(8), g.info(9), g.info(10), "stages", [g.info(1, i) for i in range(g.info(0))], "carry at load", g.info(7))
pos = 0
act = [k for k in (ks or range(M)) if synth.channel_kind(k) != "empty"]
for s in splits:
    a = g.process_block(x[pos:pos + s], want=("pcm", "resampled", "chan", "fm"))
    b = o.process_block(x[pos:pos + s], want=("pcm", "resampled", "chan", "fm"))
    pos += s
    def rel(u, v):
        return float(np.abs(u - v).max() / max(1e-30, np.abs(v).max())) if u.size else 0.0
    nr = min(len(a["resampled"]), len(b["resampled"]))
    bad = np.nonzero(np.abs(a["resampled"][:nr] - b["resampled"][:nr]) > 1e-4 * max(1e-30, np.abs(b["resampled"]).max()))[0] if nr else []
    print("block %6d: frames %d/%d resampled %d/%d rel %.2e (first bad idx %s of %d) chan %.2e fm %.2e pcm %d" % (
        s, a["n_frames"], b["n_frames"], len(a["resampled"]), len(b["resampled"]), rel(a["resampled"][:nr], b["resampled"][:nr]),
        (bad[:3].tolist(), bad[-1] if len(bad) else None), nr, rel(a["chan"][act], b["chan"][act]), rel(a["fm"][act], b["fm"][act]),
        int(np.abs(a["pcm"][act].astype(np.int32) - b["pcm"][act].astype(np.int32)).max()) if a["n_frames"] else 0), flush=True)
