#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: random configurations, options and ragged block splits; int16 PCM within +-1 LSB on
every channel that carries a signal.  Run on the GPU box: gpurun -- python3 tools/soak.py [seconds] [seed]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from sdr_pmr446_amd import chain, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CFGS = [(1.024e6, 16), (2.4e6, 16), (3.2e6, 32), (1.6e6, 32), (10.0e6, 64), (4.0e6, 64), (12.8e6, 128), (61.44e6, 256),
        (0.4e6, 16), (0.25e6, 8), (1.0e6, 4), (25.0e6, 512)]
t_end = time.time() + budget
n_cases = 0
worst = 0
while time.time() < t_end:
    fs, M = CFGS[rng.integers(len(CFGS))]
    opts = dict(lowpass=bool(rng.integers(2)) and rng.random() < 0.3, deemph_fir=rng.random() < 0.2)
    max_block = int(rng.choice([3000, 20000, 100000, 400000]))
    nblk = int(rng.integers(1, 6))
    splits = [int(rng.integers(0, max_block + 1)) if rng.random() < 0.8 else int(rng.integers(0, 40)) for _ in range(nblk)]
    n = sum(splits)
    if n * M > 6e7 or n == 0:
        continue
    ks = None if M <= 64 else list(range(0, M, M // 16))
    x = synth.synth_iq(n, fs, M, channels=ks, dev_hz=float(rng.choice([500.0, 1500.0, 2500.0])),
                       dc_offset=float(rng.choice([0.0, 0.003])))
    want = ("pcm", "rssi") if rng.random() < 0.5 else ("pcm",)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max_block, **opts)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max_block, **opts)
    pg, po, pos = [], [], 0
    for s in splits:
        a = g.process_block(x[pos:pos + s], want=want)
        b = o.process_block(x[pos:pos + s], want=want)
        assert a["n_frames"] == b["n_frames"], (fs, M, splits, a["n_frames"], b["n_frames"])
        pg.append(a["pcm"]); po.append(b["pcm"]); pos += s
        if "rssi" in want and a["n_frames"] > 8:
            act = [k for k in (ks or range(M)) if synth.channel_kind(k) != "empty"]
            assert np.abs(a["rssi"][act] - b["rssi"][act]).max() < 0.05, (fs, M, "rssi")
    pg, po = np.concatenate(pg, axis=1), np.concatenate(po, axis=1)
    act = [k for k in (ks or range(M)) if synth.channel_kind(k) != "empty"]
    d = int(np.abs(pg[act].astype(np.int32) - po[act].astype(np.int32)).max()) if pg.shape[1] else 0
    worst = max(worst, d)
    n_cases += 1
    status = "ok" if d <= 1 else "FAIL"
    print("%s fs=%g M=%d opts=%s max_block=%d splits=%s frames=%d maxdiff=%d" % (status, fs, M, opts, max_block, splits, pg.shape[1], d), flush=True)
    g.close(); o.close()
    if d > 1:
        sys.exit(1)
print("soak: %d cases, worst |pcm diff| = %d LSB" % (n_cases, worst))
