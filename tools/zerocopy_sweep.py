#!/usr/bin/env python3
"""Synchronous host call (pmr_chain_process_block_f32, pinned input, pcm + audio + rssi out) at the reference configuration:
median latency per block size.  Run once as is and once with PMR_ZEROCOPY=0 (switches are read when the handle is created); blocks
above 2^18 samples always go through the copy engines."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain, synth
sizes = [int(a) for a in sys.argv[1:]] or [25000, 100000, 1 << 18, 1 << 20, 1 << 22]
fs, M = 1.024e6, 16
x = synth.synth_iq(max(sizes), fs, M)
g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
S = g.max_frames
pcm = np.zeros((M, S), np.int16); audio = np.zeros((M, S), np.float32); rssi = np.zeros(M, np.float32); ns = C.c_uint(0)
pin = g.pinned_array(max(sizes)); pin[:] = x
print("PMR_ZEROCOPY=%s" % os.environ.get("PMR_ZEROCOPY", "(default on)"))
for n in sizes:
    def call():
        rc = g._L.pmr_chain_process_block_f32(g.h, pin.ctypes.data, n, pcm.ctypes.data, audio.ctypes.data, S, C.byref(ns), None, rssi.ctypes.data)
        assert rc == 0
    for _ in range(10): call()
    reps = 300 if n <= (1 << 18) else 60
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); call(); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    print("  %8d samples: median %8.1f us  p10 %8.1f  p99 %8.1f   %.2f GS/s" % (n, np.median(t), np.percentile(t, 10), np.percentile(t, 99), n / np.median(t) / 1e3))
