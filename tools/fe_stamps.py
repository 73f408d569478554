#!/usr/bin/env python3
"""Diagnostic: per-phase wall cycles of k_frontend (PMR_FE_STAMP=1), averaged per tile."""
import os, sys
os.environ["PMR_FE_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdr_pmr446_amd import chain as pmr
from sdr_pmr446_amd.synth_torch import synth_iq_torch
fs, M, lb = 2.4e6, 16, 26
if len(sys.argv) > 1 and sys.argv[1] == "cfg5": fs, M = 1e9, 1024
if len(sys.argv) > 1 and sys.argv[1] == "cfg3": fs, M = 61.44e6, 256
block = 1 << lb
ch = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=block)
iq = torch.randn(block, 2, device="cuda") * 0.2
pcm = torch.zeros((M, ch.max_frames), dtype=torch.int16, device="cuda")
for _ in range(3):
    ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
ch.synchronize()
st = ch.debug_read(2, np.uint64)
n = max(int(st[4]), 1)
print("tiles", n, "cycles/tile  A(load) %.0f  B(dc) %.0f  C(cascade) %.0f  D(resamp) %.0f  total %.0f" %
      (st[0] / n, st[1] / n, st[2] / n, st[3] / n, sum(st[:4]) / n))
