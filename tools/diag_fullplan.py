#!/usr/bin/env python3
"""Where does GPU-vs-oracle PCM error come from when EVERY channel of a wide configuration carries a signal?
Per-stage error statistics (resampled, chan, fm, audio, pcm) for the sparse test plan and the full SURVEY plan.
Run on the GPU box: gpurun -- python3 tools/diag_fullplan.py [cfg5|cfg3] [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from sdr_pmr446_amd import chain, synth
import torch
from sdr_pmr446_amd.synth_torch import synth_iq_torch

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
fs, M = {"cfg5": (1.0e9, 1024), "cfg3": (61.44e6, 256), "cfg2": (2.4e6, 16)}[name]
n = 1 << lg
WANT = ("pcm", "audio", "chan", "fm", "resampled")
for plan, ks in (("sparse", list(range(0, M, 73 if M == 1024 else 5))), ("full", None)):
    x = synth_iq_torch(n, fs, M, torch.device('cuda', 0), channels=ks, dev_hz=1500.0).cpu().numpy()
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    ro, rg = o.process_block(x, want=WANT), g.process_block(x, want=WANT)
    act = [k for k in (ks or range(M)) if synth.channel_kind(k) != "empty"]
    print("==", name, plan, "rms(x)=%.4f" % np.sqrt(np.mean(np.abs(x) ** 2)), "frames", ro["n_frames"])
    rs = np.abs(rg["resampled"] - ro["resampled"]); print("resampled: max %.3g rms %.3g  signal rms %.3g" % (rs.max(), np.sqrt(np.mean(rs**2)), np.sqrt(np.mean(np.abs(ro["resampled"])**2))))
    c = np.abs(rg["chan"][act] - ro["chan"][act]); print("chan     : max %.3g rms %.3g  signal rms %.3g" % (c.max(), np.sqrt(np.mean(c**2)), np.sqrt(np.mean(np.abs(ro["chan"][act])**2))))
    sk = 60
    f = np.abs(rg["fm"][act][:, sk:] - ro["fm"][act][:, sk:]); print("fm       : max %.3g rms %.3g" % (f.max(), np.sqrt(np.mean(f**2))))
    st = 700
    if ro["n_frames"] > st + 10:
        a = np.abs(rg["audio"][act][:, st:] - ro["audio"][act][:, st:]); print("audio    : max %.3g rms %.3g  (1 LSB = %.3g)" % (a.max(), np.sqrt(np.mean(a**2)), 1 / 32767))
    d = np.abs(rg["pcm"][act].astype(np.int32) - ro["pcm"][act].astype(np.int32))
    print("pcm      : max %d  hist" % d.max(), np.bincount(d.ravel())[:6], " min |chan| over act:", float(np.abs(ro["chan"][act][:, sk:]).min()))
    if d.max() > 1:
        kk, tt = np.nonzero(d > 1)
        print("   >1 LSB at (channel kind frame |chan|):", [(act[k], synth.channel_kind(act[k]), int(t), float(np.abs(ro["chan"][act[k], t]))) for k, t in list(zip(kk, tt))[:8]])
    mc = np.abs(ro["chan"][act][:, sk:]).mean(axis=1)
    med = np.median(mc)
    low = [(act[i], round(float(mc[i] / med), 3)) for i in np.argsort(mc)[:10]]
    print("   lowest mean|chan| / median:", low)
    worst = np.argsort(-d.max(axis=1))[:10]
    print("   worst channels (k, max diff, mean|chan|/median):", [(act[i], int(d[i].max()), round(float(mc[i] / med), 3)) for i in worst])
    o.close(); g.close()
