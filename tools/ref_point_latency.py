#!/usr/bin/env python3
"""Latency of one synchronous pmr_chain_process_block call at the REFERENCE's operating point (1.024 MS/s, 16 channels,
100000-sample blocks = 97.7 ms of signal, host buffers in and out), and of pmr_dsd_process_block (200000 samples)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain, synth
x = synth.synth_iq(100000, 1.024e6, 16)
g = chain.PmrChain()
S = g.max_frames
pcm = np.zeros((16, S), np.int16); audio = np.zeros((16, S), np.float32); rssi = np.zeros(16, np.float32); ns = C.c_uint(0)
def call():
    rc = g._L.pmr_chain_process_block_f32(g.h, x.ctypes.data, len(x), pcm.ctypes.data, audio.ctypes.data, S, C.byref(ns), None, rssi.ctypes.data)
    assert rc == 0
for _ in range(20): call()
t = []
for _ in range(200):
    t0 = time.perf_counter(); call(); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
print("pmr_chain_process_block_f32, 100000 samples (97.7 ms of signal): median %.0f us, p99 %.0f us -> %.0fx real time" %
      (np.median(t), np.percentile(t, 99), 97656.0 / np.median(t)))
d = chain.PmrDsd()
xd = synth.synth_iq(200000, 1.024e6, 1)
out = np.zeros(d.max_out, np.int16); nz = C.c_uint(0)
def calld():
    rc = d._L.pmr_dsd_process_block(d.h, xd.ctypes.data, len(xd), out.ctypes.data, None, d.max_out, C.byref(nz))
    assert rc == 0
for _ in range(20): calld()
t = []
for _ in range(200):
    t0 = time.perf_counter(); calld(); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
print("pmr_dsd_process_block, 200000 samples (195.3 ms of signal): median %.0f us, p99 %.0f us" % (np.median(t), np.percentile(t, 99)))
