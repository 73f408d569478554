#!/usr/bin/env python3
"""Diagnostic: isolated k_frontend time with phases ablated (PMR_FE_ABLATE bits: 1 load, 2 dc scan, 4 cascade,
8 resampler).  Results are wrong under ablation by construction; only the timings mean anything."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_pmr446_amd import chain as pmr
fs, M, lb = 2.4e6, 16, 26
if len(sys.argv) > 1 and sys.argv[1] == "cfg5": fs, M = 1e9, 1024
if len(sys.argv) > 1 and sys.argv[1] == "cfg3": fs, M = 61.44e6, 256
block = 1 << lb
ch = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=block)
ch.set_overlap(False)
iq = torch.randn(block, 2, device="cuda") * 0.2
pcm = torch.zeros((M, ch.max_frames), dtype=torch.int16, device="cuda")
for ab in (0, 1, 2, 4, 8, 6, 14, 15, 13, 11, 7, 0):
    os.environ["PMR_FE_ABLATE"] = str(ab)
    for _ in range(2):
        ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
    ch.synchronize(); ch.profile_reset(); ch.profile_enable(1)
    for _ in range(5):
        ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=ch.max_frames)
    ch.synchronize(); ch.profile_enable(0)
    p = ch.profile()
    ms, n = p["k_frontend"]
    print("ablate %2d (skip:%s%s%s%s)  k_frontend %.4f ms" % (ab, " load" if ab & 1 else "", " dc" if ab & 2 else "",
          " cascade" if ab & 4 else "", " resamp" if ab & 8 else "", ms / n), flush=True)
