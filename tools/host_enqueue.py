#!/usr/bin/env python3
"""Diagnostic: host enqueue time per process_block_device call vs GPU time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_pmr446_amd import chain as pmr
fs, M, block = 2.4e6, 16, 1 << 26
ch = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=block)
iq = torch.randn(block, 2, device="cuda") * 0.2
pcm = torch.zeros((M, ch.max_frames), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
ch.synchronize()
N = 40
t0 = time.perf_counter()
marks = []
for _ in range(N):
    ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
    marks.append(time.perf_counter())
t1 = time.perf_counter()
ch.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/call (first 5: %s) ; total %.1f us/step" % ((t1 - t0) / N * 1e6,
      " ".join("%.0f" % ((marks[i] - (marks[i - 1] if i else t0)) * 1e6) for i in range(5)), (t2 - t0) / N * 1e6))
