#!/usr/bin/env python3
"""Throughput of the `dsd_in` chain (include/pmr_dsd.h; reference src/dsd_in.c:160-178) on one GPU: 1.024 MS/s
stream, 2^26-sample blocks resident in HBM (rotating through four distinct blocks), s16le output left in HBM; the CPU oracle timed
beside it.  One JSON line with a `roofline` object: algorithmic bytes = 8 B in + 2 B x 48000 / fs_in out per input sample, over
the WHOLE step (the chain's front end is the two-level form of bench.py's cfg5: level 1 is the step but for ~1/16 of the data)."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from sdr_pmr446_amd import chain

fs, lb, steps, regions, rot = 1.024e6, 26, 20, 9, 4
block = 1 << lb
d = chain.PmrDsd(fs_in=fs, max_block=block)
iq = chain.synth_iq_device(rot * block, fs, 1, period_log2=lb + 2)
pcm = chain.DeviceBuffer(d.max_out * 2)
chain.device_synchronize()
pos = 0
for _ in range(3):
    d.process_block_device(iq.ptr + (pos % rot) * block * 8, block, pcm.ptr, None, d.max_out); pos += 1
d.synchronize()
dts = []
for _ in range(regions):
    t0 = time.perf_counter()
    for _ in range(steps):
        nz = d.process_block_device(iq.ptr + (pos % rot) * block * 8, block, pcm.ptr, None, d.max_out); pos += 1
    d.synchronize()
    dts.append(time.perf_counter() - t0)
dt = statistics.median(dts)
# parity of the timed path on the first 2^20 samples of a fresh stream (un-synchronised device call vs the oracle)
d.reset()
n_chk = 1 << 20
nz_chk = d.process_block_device(iq.ptr, n_chk, pcm.ptr, None, d.max_out)
d.synchronize()
got = pcm.download(np.int16, nz_chk).astype(np.int32)
o = oracle.OracleDsd(fs_in=fs, max_block=n_chk)
x = iq.download(np.complex64, n_chk)
ref = o.process_block(x)["pcm"].astype(np.int32)
ok = len(ref) == len(got) and int(np.abs(got - ref).max()) <= 1
n, t1 = 0, time.perf_counter()
while time.perf_counter() - t1 < 6.0:
    o.process_block(x); n += 1
cpu = n * n_chk / (time.perf_counter() - t1) / 1e6
b_alg = 8.0 + 2.0 * 48000.0 / fs
ach = b_alg * block / (dt / steps) / 1e9
print(json.dumps({"metric": "complex-IQ Msamples/s through the dsd_in chain", "value": steps * block / dt / 1e6, "unit": "Msamples/s",
                  "ms_per_step": dt / steps * 1e3, "block_samples": block, "out_samples_per_step": nz, "regions": regions,
                  "ms_per_step_min_max": [min(dts) / steps * 1e3, max(dts) / steps * 1e3],
                  "input": "%d distinct device-resident blocks rotated through the steps" % rot,
                  "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                               "algorithmic_bytes_per_sample": b_alg,
                               "note": "whole-step time (no per-kernel events on this handle); level 1 of the shared two-level front end is the step"},
                  "parity_checked": {"ok": bool(ok), "max_abs_pcm_diff_lsb": int(np.abs(got - ref).max()) if len(ref) == len(got) else -1,
                                     "samples": int(len(got)), "mode": "un-synchronised device call on 2^20 samples vs oracle.OracleDsd"},
                  "cpu_baseline": {"value": cpu, "unit": "Msamples/s", "cores": 1, "kind": "port"}}))
sys.exit(0 if ok else 1)
