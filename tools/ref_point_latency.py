#!/usr/bin/env python3
"""Host-buffer entry points at the REFERENCE's operating point (1.024 MS/s, 16 channels, 100000-sample blocks = 97.7 ms of
signal): latency of one synchronous pmr_chain_process_block_f32 call, steady-state time per block of the asynchronous
submit / collect pair, and pmr_dsd_process_block (200000 samples)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain, synth
x = synth.synth_iq(100000, 1.024e6, 16)
g = chain.PmrChain()
S = g.max_frames
pcm = np.zeros((16, S), np.int16); audio = np.zeros((16, S), np.float32); rssi = np.zeros(16, np.float32); ns = C.c_uint(0)
for kind in (() if "--async-only" in sys.argv else ("pageable", "pinned")):
    xi = x
    if kind == "pinned":
        xi = g.pinned_array(len(x)); xi[:] = x
    def call():
        rc = g._L.pmr_chain_process_block_f32(g.h, xi.ctypes.data, len(xi), pcm.ctypes.data, audio.ctypes.data, S, C.byref(ns), None, rssi.ctypes.data)
        assert rc == 0
    for _ in range(20): call()
    t = []
    for _ in range(300):
        t0 = time.perf_counter(); call(); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    print("pmr_chain_process_block_f32 (%s input), 100000 samples (97.7 ms of signal): median %.0f us, p99 %.0f us -> %.0fx real time" %
          (kind, np.median(t), np.percentile(t, 99), 97656.0 / np.median(t)))
# the receiver's own sample formats, read in place and converted by the front end (pmr_chain_process_block_fmt)
xi2 = np.empty(2 * len(x), np.float32); xi2[0::2] = x.real; xi2[1::2] = x.imag
for name, code, dt_np, conv in (("uint8 (rtl_sdr)", 2, np.uint8, lambda v: np.clip(np.round(v * 127.5 + 127.5), 0, 255)),
                                ("int16", 1, np.int16, lambda v: np.clip(np.round(v * 32768.0), -32768, 32767))):
    raw = g.pinned_array(2 * len(x), dt_np); raw[:] = conv(xi2).astype(dt_np)
    def callf():
        rc = g._L.pmr_chain_process_block_fmt(g.h, raw.ctypes.data, code, len(x), pcm.ctypes.data, audio.ctypes.data, S, C.byref(ns), None, rssi.ctypes.data)
        assert rc == 0
    for _ in range(20): callf()
    t = []
    for _ in range(300):
        t0 = time.perf_counter(); callf(); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    print("pmr_chain_process_block_fmt (%s, pinned input read in place), 100000 samples: median %.0f us, p99 %.0f us -> %.0fx real time" %
          (name, np.median(t), np.percentile(t, 99), 97656.0 / np.median(t)))
if "--sync-only" in sys.argv:
    g.close()
    raise SystemExit(0)
# asynchronous pair, pipe kept full
depth = g._L.pmr_chain_max_in_flight(g.h)
bufs = [g.pinned_array(len(x)) for _ in range(depth)]
for b in bufs: b[:] = x
def sub(i): assert g._L.pmr_chain_submit_block(g.h, bufs[i % depth].ctypes.data, len(x), 1 | 2 | 4) == 0
def col(): assert g._L.pmr_chain_collect_block(g.h, pcm.ctypes.data, audio.ctypes.data, S, C.byref(ns), None, rssi.ctypes.data) == 0
g.reset()
for i in range(depth): sub(i)
N = 600
t0 = time.perf_counter()
for i in range(N):
    col(); sub(i)
dt = time.perf_counter() - t0
for i in range(depth): col()
print("pmr_chain_submit_block / collect_block, %d blocks in flight: %.0f us per 100000-sample block (%.1f MS/s, %.0fx real time)" %
      (depth, dt / N * 1e6, N * 1e5 / dt / 1e6, 97656.0 / (dt / N * 1e6)))
if "--async-only" in sys.argv:
    g.close()
    raise SystemExit(0)
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
