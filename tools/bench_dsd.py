#!/usr/bin/env python3
"""Throughput of the `dsd_in` chain (include/pmr_dsd.h) on one GPU: 1.024 MS/s stream, 2^26-sample
blocks resident in HBM, s16le output left in HBM; the CPU oracle timed beside it.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from sdr_pmr446_amd import chain
from sdr_pmr446_amd.synth_torch import synth_iq_torch

fs, block, steps = 1.024e6, 1 << 26, 20
d = chain.PmrDsd(fs_in=fs, max_block=block)
iq = synth_iq_torch(block, fs, 1, torch.device("cuda", 0))
pcm = torch.zeros(d.max_out, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    d.process_block_device(iq.data_ptr(), block, pcm.data_ptr(), None, d.max_out)
d.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    nz = d.process_block_device(iq.data_ptr(), block, pcm.data_ptr(), None, d.max_out)
d.synchronize()
dt = time.perf_counter() - t0
o = oracle.OracleDsd(fs_in=fs, max_block=1 << 20)
x = iq[:1 << 20].cpu().numpy()
o.process_block(x)
n, t1 = 0, time.perf_counter()
while time.perf_counter() - t1 < 6.0:
    o.process_block(x); n += 1
cpu = n * (1 << 20) / (time.perf_counter() - t1) / 1e6
print(json.dumps({"metric": "complex-IQ Msamples/s through the dsd_in chain", "value": steps * block / dt / 1e6, "unit": "Msamples/s",
                  "ms_per_step": dt / steps * 1e3, "block_samples": block, "out_samples_per_step": nz,
                  "hbm_GBps_input": steps * block * 8 / dt / 1e9, "cpu_baseline": {"value": cpu, "unit": "Msamples/s", "cores": 1, "kind": "port"}}))
