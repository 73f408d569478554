#!/usr/bin/env python3
"""Diagnostic: isolated k_fir_mfma16 time with parts ablated (PMR_FIR_ABLATE bits: 1 staging loads, 2 half the MFMA
steps, 4 no epilogue).  Results are wrong under ablation; only the timings mean anything."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_pmr446_amd import chain as pmr
fs, M, lb = 2.4e6, 16, 26
block = 1 << lb
ch = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=block)
ch.set_overlap(False)
iq = torch.randn(block, 2, device="cuda") * 0.2
pcm = torch.zeros((M, ch.max_frames), dtype=torch.int16, device="cuda")
for ab in (0, 1, 2, 4, 3, 7, 0):
    os.environ["PMR_FIR_ABLATE"] = str(ab)
    for _ in range(2):
        ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
    ch.synchronize(); ch.profile_reset(); ch.profile_enable(1)
    for _ in range(5):
        ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
    ch.synchronize(); ch.profile_enable(0)
    p = ch.profile()
    for k, (ms, n) in p.items():
        if "fir" in k:
            print("ablate %d  %s %.4f ms" % (ab, k, ms / n), flush=True)
