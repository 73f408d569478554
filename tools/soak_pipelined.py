#!/usr/bin/env python3
"""Pipelined soak (GPU box): random configurations and random ragged block sequences through pmr_chain_process_block_device with
NOTHING synchronised in between (three blocks in flight, both streams busy), then the same sequence with a synchronise after
every block -- PCM, CTCSS events and the waterfall PSD must agree bit for bit.  No oracle involved: this looks for ordering /
ring-reuse / event hazards, not for arithmetic.     usage: soak_pipelined.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain as pmr

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CFGS = [(2.4e6, 16, 22), (61.44e6, 256, 22), (1.0e9, 1024, 24), (1.024e6, 16, 20), (9.6e6, 64, 21)]
t_end = time.time() + budget
cases = 0
tot_samples = tot_frames = 0
while time.time() < t_end:
    fs, M, lb = CFGS[rng.integers(len(CFGS))]
    maxb = 1 << lb
    nblk = int(rng.integers(8, 24))
    sizes = []
    for _ in range(nblk):
        k = rng.integers(6)
        sizes.append(0 if k == 0 else int(rng.integers(1, 64)) if k == 1 else maxb if k == 2 else int(rng.integers(1, maxb + 1)))
    total = sum(sizes)
    ctcss, mask, spec = bool(rng.integers(2)), bool(rng.integers(3) == 0), bool(rng.integers(3) == 0)
    iq = pmr.synth_iq_device(max(total, 1), fs, M, stream_id=int(rng.integers(1000)), channel_step=max(1, M // 16))
    g = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=maxb)
    S = g.max_frames
    outs = [pmr.DeviceBuffer(M * S * 2) for _ in range(nblk)]
    if ctcss:
        g._check(g._L.pmr_chain_ctcss_enable(g.h, 1))
    if spec:
        g.spectrum_enable(64)
    mask0 = [int(c) for c in rng.choice(M, size=min(M, 3), replace=False)]

    # the open-channel set changes in mid-stream in some cases (the carry pass of the one-level front end then changes streams)
    toggles = {int(b): ([int(c) for c in rng.choice(M, size=min(M, int(rng.integers(1, 4))), replace=False)] if rng.integers(2) else None)
               for b in rng.choice(nblk, size=int(rng.integers(0, 4)), replace=False)}

    def run(sync_each):
        pos, ns, extra = 0, [], []
        if mask:
            g.set_channel_mask(mask0)
        else:
            g.set_channel_mask(None)
        for b, n in enumerate(sizes):
            if b in toggles:
                g.set_channel_mask(toggles[b])
            ns.append(g.process_block_device(iq.ptr + pos * 8, n, d_pcm=outs[b].ptr, stride=S))
            pos += n
            if sync_each:
                g.synchronize()
        g.synchronize()
        pcm = [outs[b].download(np.int16, M * S).reshape(M, S)[:, :ns[b]].copy() for b in range(nblk)]
        ev = g.ctcss_read() if ctcss else None
        psd = g.spectrum_read() if spec else None
        return ns, pcm, ev, psd

    for o in outs:
        o.upload(np.zeros(M * S, np.int16))                   # rows of closed channels are left untouched by the chain
    pmr.device_synchronize()
    a = run(False)
    g.reset()
    for o in outs:
        o.upload(np.zeros(M * S, np.int16))
    b = run(True)
    ok = a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    if ctcss:
        ok = ok and a[2].shape == b[2].shape and a[2].tobytes() == b[2].tobytes()
    if spec:
        ok = ok and a[3][1] == b[3][1] and np.array_equal(a[3][0], b[3][0])
    if not ok:
        print("soak_pipelined: MISMATCH fs=%g M=%d sizes=%s ctcss=%d mask=%d spec=%d" % (fs, M, sizes, ctcss, mask, spec))
        sys.exit(1)
    cases += 1
    tot_samples += total; tot_frames += sum(a[0])
    g.close(); iq.free()
    for o in outs:
        o.free()
print("soak_pipelined: %d cases (%.3g samples, %d frames per channel), all bit-identical between un-synchronised and per-block-synchronised runs" % (cases, tot_samples, tot_frames))
