#!/usr/bin/env python3
"""GPU box: the 24 (configuration, stream id) cases of tests/test_gpu_streams.py, one line each -- how many discriminator samples are
ill-conditioned after the reset, how far the chain's discriminator is from the oracle's there, how many PCM samples that reaches, the
largest PCM difference among them, and what of it the audio filter's response to the measured discriminator difference leaves
UNEXPLAINED (sdr_pmr446_amd/parity_rule.py).  -> profiles/r06_stream_parity.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from sdr_pmr446_amd import chain, parity_rule, synth

CFG = {"cfg5": (1.0e9, 1024, 25, 2), "cfg3": (61.44e6, 256, 23, 2), "cfg2": (2.4e6, 16, 20, 3)}
hp, b0, b1, a1 = parity_rule.fixtures(ROOT)
h = parity_rule.audio_response(hp, 4.0, b0, b1, a1)
print("%-6s %3s | %9s %5s %11s | %9s %9s %9s %11s | %6s %s" % ("cfg", "sid", "ill fm", "last", "max |dfm|", "pcm reach", "> 1 LSB", "max |d|", "unexplained", "strict", "ok"))
for name, (fs, M, lb, nblk) in CFG.items():
    for sid in range(8):
        block = 1 << lb
        iq = chain.synth_iq_device(nblk * block, fs, M, stream_id=sid, period_log2=28)
        x = iq.download(np.complex64, nblk * block)
        g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
        S = g.max_frames
        bufs = [chain.DeviceBuffer(M * S * 2) for _ in range(nblk)]
        ns = [g.process_block_device(iq.ptr + b * block * 8, block, d_pcm=bufs[b].ptr, stride=S) for b in range(nblk)]
        g.synchronize()
        got = np.concatenate([bufs[b].download(np.int16, M * S).reshape(M, S)[:, :ns[b]] for b in range(nblk)], axis=1)
        g.close(); iq.free()
        for b in bufs:
            b.free()
        nfirst = min(block, 1 << 22)
        gd = chain.PmrChain(fs_in=fs, num_channels=M, max_block=nfirst)
        rd = gd.process_block(x[:nfirst], want=("fm",)); gd.close()
        o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=1 << 22)
        outs = [o.process_block(x[p:p + (1 << 22)], want=("pcm", "chan", "fm")) for p in range(0, nblk * block, 1 << 22)]
        o.close()
        ref = np.concatenate([r["pcm"] for r in outs], axis=1); chan = np.concatenate([r["chan"] for r in outs], axis=1)
        act = synth.signal_channels(M, fs)
        F = rd["n_frames"]
        v = parity_rule.check(got[act], ref[act], chan[act], rd["fm"][act], outs[0]["fm"][act, :F], h)
        i = v["ill_conditioned"]
        print("%-6s %3d | %9d %5d %11.3g | %9d %9d %9d %11.2f | %6d %s" % (name, sid, i["discriminator_samples"], i["last_frame"], i["max_abs_discriminator_diff"],
              i["pcm_samples_reached"], i["samples_over_1_lsb"], i["max_abs_pcm_diff_lsb_reached"], i["max_abs_unexplained_lsb"], v["max_abs_pcm_diff_lsb"], v["ok"]), flush=True)
