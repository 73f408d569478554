#!/usr/bin/env python3
"""GPU box: where does one PCM sample of one channel part from the oracle?  python3 tools/diag_stream_detail.py cfg5 1 410 200"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMR_NO_TORCH"] = "1"
import numpy as np
import oracle
from sdr_pmr446_amd import chain
W = {"cfg2": (2.4e6, 16), "cfg3": (61.44e6, 256), "cfg5": (1.0e9, 1024)}
name, sid, k, f = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fs, M = W[name]
lb = 26
block = 1 << lb
iq = chain.synth_iq_device(block, fs, M, stream_id=sid, period_log2=lb + 2)
x = iq.download(np.complex64, block)
res = {}
for form in (None, "direct"):
    if form:
        os.environ["PMR_FIR"] = form
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
    os.environ.pop("PMR_FIR", None)
    res[form] = g.process_block(x, want=("pcm", "audio", "fm", "chan"))
    g.close()
o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=block)
ro = o.process_block(x, want=("pcm", "audio", "fm", "chan"))
sl = slice(max(0, f - 4), f + 5)
for form in (None, "direct"):
    r = res[form]
    print("== GPU form", form or "fft")
    print(" pcm  gpu", r["pcm"][k, sl].tolist()); print(" pcm  orc", ro["pcm"][k, sl].tolist())
    print(" audio diff", (r["audio"][k, sl] - ro["audio"][k, sl]).tolist())
    dfm = np.abs(r["fm"][k] - ro["fm"][k])
    print(" fm: max |diff| over the row %.3e at frame %d; around f: %s" % (dfm.max(), int(dfm.argmax()), dfm[sl].tolist()))
    dch = np.abs(r["chan"][k] - ro["chan"][k])
    print(" chan: max |diff| %.3e (scale %.3e); |chan| around the worst fm frame: %s" % (dch.max(), np.abs(ro["chan"][k]).max(), np.abs(ro["chan"][k][max(0, int(dfm.argmax()) - 2):int(dfm.argmax()) + 3]).tolist()))
    da = np.abs(r["audio"][k] - ro["audio"][k])
    print(" audio: max |diff| %.3e at frame %d, scale %.3f" % (da.max(), int(da.argmax()), np.abs(ro["audio"][k]).max()))
    print(" audio*32767 at f: gpu %.4f orc %.4f" % (r["audio"][k, f] * 32767.0, ro["audio"][k, f] * 32767.0))
