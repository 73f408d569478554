#!/usr/bin/env python3
"""GPU box: per-channel PCM deviation from the oracle on one bench block of stream `sid` (what a rank of bench.py --gpus N checks).
   python3 tools/diag_stream_parity.py [workload cfg5] [stream_id 1] [log2_block 26]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMR_NO_TORCH"] = "1"
import numpy as np
import oracle
from sdr_pmr446_amd import chain, synth
W = {"cfg2": (2.4e6, 16), "cfg3": (61.44e6, 256), "cfg5": (1.0e9, 1024)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
sid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lb = int(sys.argv[3]) if len(sys.argv) > 3 else 26
fs, M = W[name]
block = 1 << lb
iq = chain.synth_iq_device(block, fs, M, stream_id=sid, period_log2=lb + 2)
x = iq.download(np.complex64, block)
g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
S = g.max_frames
pcm = chain.DeviceBuffer(M * S * 2)
ns = g.process_block_device(iq.ptr, block, d_pcm=pcm.ptr, stride=S)
g.synchronize()
got = pcm.download(np.int16, M * S).reshape(M, S)[:, :ns].astype(np.int32)
o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=1 << 22)
ref = np.concatenate([o.process_block(x[p:p + (1 << 22)], want=("pcm",))["pcm"] for p in range(0, block, 1 << 22)], axis=1).astype(np.int32)
d = np.abs(got - ref)
act = synth.signal_channels(M, fs)
print("stream", sid, name, "frames", ns, "max over signal channels", int(d[act].max()), " over all", int(d.max()))
bad = [k for k in act if d[k].max() > 1]
for k in bad[:20]:
    idx = np.flatnonzero(d[k] > 1)
    print("  channel %d (%s): %d samples > 1 LSB (max %d) at frames %s; |pcm| there %s, rms of the row %.0f; offset from band centre %.1f kHz" % (
        k, synth.channel_kind(k), idx.size, d[k].max(), idx[:6].tolist(), np.abs(ref[k][idx[:6]]).tolist(), float(np.sqrt((ref[k] ** 2.0).mean())),
        (k - (M - 1) / 2) * 12.5))
print("channels with > 1 LSB:", len(bad), "of", len(act))
T = 26 + 383
print("  start-up frames [0, %d): max %d, within 1 LSB %.6f;  frames >= %d: max %d" % (T, int(d[act][:, :T].max()), float((d[act][:, :T] <= 1).mean()), T,
      int(d[act][:, T:].max()) if d.shape[1] > T else -1))
