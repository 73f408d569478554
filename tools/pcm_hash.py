#!/usr/bin/env python3
"""GPU box: sha256 of what the chain produces on fixed synthetic streams (PCM of three ragged un-synchronised device calls per workload, and
the resampled stream of a synchronous call) -- to show that two BUILDS of the library are bit-identical (same operations in the same
order):  python3 tools/pcm_hash.py ; PMR_LIBRARY=build_ab/X/libpmr446_hip.so python3 tools/pcm_hash.py"""
import hashlib, os, sys
os.environ.setdefault("PMR_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain
WORK = {"ref": (1.024e6, 16, 18), "cfg2": (2.4e6, 16, 22), "cfg3": (61.44e6, 256, 24), "cfg5": (1.0e9, 1024, 26)}
for name in (sys.argv[1:] or list(WORK)):
    fs, M, lb = WORK[name]
    n = 1 << lb
    splits = [n // 2 + 4321, n // 4 - 4321 - 7, n // 4 + 7]
    iq = chain.synth_iq_device(n, fs, M, stream_id=2)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(splits))
    S = g.max_frames
    h = hashlib.sha256()
    pos = 0
    bufs = [chain.DeviceBuffer(M * S * 2) for _ in splits]
    ns = []
    for b, k in zip(bufs, splits):
        ns.append(g.process_block_device(iq.ptr + pos * 8, k, d_pcm=b.ptr, stride=S)); pos += k
    g.synchronize()
    for b, k in zip(bufs, ns):
        h.update(np.ascontiguousarray(b.download(np.int16, M * S).reshape(M, S)[:, :k]).tobytes())
    g.close()
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=min(n, 1 << 20))
    x = iq.download(np.complex64, min(n, 1 << 20))
    r = g.process_block(x, want=("resampled",))
    hr = hashlib.sha256(r["resampled"].tobytes()).hexdigest()[:16]
    g.close(); iq.free()
    for b in bufs:
        b.free()
    print("%-5s frames %s  pcm %s  resampled %s" % (name, ns, h.hexdigest()[:16], hr), flush=True)
