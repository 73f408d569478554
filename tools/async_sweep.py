#!/usr/bin/env python3
"""Host-fed rates per block size at cfg5: synchronous call vs the asynchronous submit / collect pair, cf32 input from pinned memory."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain as pmr
fs, M = 1.0e9, 1024
sizes = [int(a) for a in sys.argv[1:]] or [1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23]
g = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
S = g.max_frames
L = g._L
depth = L.pmr_chain_max_in_flight(g.h)
pinned = [g.pinned_array(max(sizes)) for _ in range(depth)]
x = (np.random.default_rng(1).standard_normal(2 * max(sizes)).astype(np.float32) * 0.1).view(np.complex64)
for p in pinned: p[:] = x
pcm = np.zeros((M, S), np.int16); ns = C.c_uint(0)
for nb in sizes:
    g.reset()
    def sync_call():
        assert L.pmr_chain_process_block(g.h, pinned[0].ctypes.data, nb, pcm.ctypes.data, S, C.byref(ns), None, None) == 0
    sync_call()
    n_it = max(8, (1 << 26) // nb)
    t0 = time.perf_counter()
    for _ in range(n_it): sync_call()
    r_sync = n_it * nb / (time.perf_counter() - t0) / 1e9
    g.reset()
    for i in range(depth): assert L.pmr_chain_submit_block(g.h, pinned[i].ctypes.data, nb, 1) == 0
    t0 = time.perf_counter()
    for i in range(n_it):
        assert L.pmr_chain_collect_block(g.h, pcm.ctypes.data, None, S, C.byref(ns), None, None) == 0
        assert L.pmr_chain_submit_block(g.h, pinned[i % depth].ctypes.data, nb, 1) == 0
    dt = time.perf_counter() - t0
    for i in range(depth): assert L.pmr_chain_collect_block(g.h, pcm.ctypes.data, None, S, C.byref(ns), None, None) == 0
    print("%9d samples (%5.1f MB): sync %.2f GS/s (%.1f GB/s)   async %.2f GS/s (%.1f GB/s)" % (nb, nb * 8 / 1e6, r_sync, r_sync * 8, n_it * nb / dt / 1e9, n_it * nb / dt / 1e9 * 8))
