#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: random configurations, options and ragged block splits; int16 PCM within +-1 LSB on
every channel that carries a signal.  Run on the GPU box: gpurun -- python3 tools/soak.py [seconds] [seed]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from sdr_pmr446_amd import chain, parity_rule, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP, B0, B1, A1 = parity_rule.fixtures(ROOT)
DE_FIR, LP = parity_rule.option_taps(ROOT)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CFGS = [(1.024e6, 16), (2.4e6, 16), (3.2e6, 32), (1.6e6, 32), (10.0e6, 64), (4.0e6, 64), (12.8e6, 128), (61.44e6, 256),
        (0.4e6, 16), (0.25e6, 8), (1.0e6, 4), (25.0e6, 512)]
t_end = time.time() + budget
n_cases = 0
worst = 0
def dsd_case():
    """dsd_in chain (include/pmr_dsd.h) against the oracle: random rates and ragged splits."""
    fs = float(rng.choice([1.024e6, 2.4e6, 250e3, 500e3, 4.0e6]))
    au = float(rng.choice([48000.0, 24000.0, 96000.0, 12500.0, 44100.0]))
    mb = int(rng.choice([5000, 60000, 200000]))
    splits = [int(rng.integers(0, mb + 1)) if rng.random() < 0.8 else int(rng.integers(0, 70)) for _ in range(int(rng.integers(1, 5)))]
    n = sum(splits)
    if n == 0:
        return 0
    x = synth.synth_iq(n, fs, 1, dev_hz=float(rng.choice([1500.0, 2500.0])), dc_offset=float(rng.choice([0.0, 0.002])))
    g = chain.PmrDsd(fs_in=fs, audio_rate=au, max_block=mb)
    o = oracle.OracleDsd(fs_in=fs, audio_rate=au, max_block=mb)
    pg, po, pos = [], [], 0
    for sp in splits:
        a = g.process_block(x[pos:pos + sp]); b = o.process_block(x[pos:pos + sp]); pos += sp
        assert a["n_out"] == b["n_out"], ("dsd", fs, au, splits)
        pg.append(a["pcm"]); po.append(b["pcm"])
    pg, po = np.concatenate(pg), np.concatenate(po)
    d = int(np.abs(pg.astype(np.int32) - po.astype(np.int32)).max()) if len(pg) else 0
    print("%s dsd fs=%g audio=%g max_block=%d splits=%s out=%d maxdiff=%d" % ("ok" if d <= 1 else "FAIL", fs, au, mb, splits, len(pg), d), flush=True)
    g.close(); o.close()
    if d > 1:
        sys.exit(1)
    return d


while time.time() < t_end:
    if rng.random() < 0.15:
        worst = max(worst, dsd_case()); n_cases += 1
        continue
    fs, M = CFGS[rng.integers(len(CFGS))]
    opts = dict(lowpass=bool(rng.integers(2)) and rng.random() < 0.3, deemph_fir=rng.random() < 0.2)
    # (round 6) another stop-band of the resampler = another pair of long half-band stages: the specialised front end is built for
    # every pair a 50 ... 72 dB design yields, 75 takes the generic kernels (M >= 16: with fewer channels the outermost ones sit on the
    # resampler's transition band)
    if M >= 16 and rng.random() < 0.5:
        opts["resamp_As"] = float(rng.choice([50.0, 55.0, 65.0, 68.0, 70.0, 75.0]))
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
    # Every case starts at a reset, so the first frames' discriminator samples can be ill-conditioned (arg() of a channel output that is
    # numerically zero while the polyphase windows fill): the PCM is held to sdr_pmr446_amd/parity_rule.py -- +-1 LSB wherever no such
    # sample reaches, and oracle + what the MEASURED discriminator difference at those samples explains through the audio path
    # elsewhere (round 6: with random stop-bands one case in a few hundred has a +pi / -pi pair there, PCM hundreds of LSB apart).
    pg, po, co, fg, fo, pos, seen = [], [], [], [], [], 0, 0
    for s in splits:
        tap = ("fm",) if seen < 64 else ()                     # the chain's discriminator output through the debug tap, start-up only
        a = g.process_block(x[pos:pos + s], want=want + tap)
        b = o.process_block(x[pos:pos + s], want=want + ("chan",) + tap)
        assert a["n_frames"] == b["n_frames"], (fs, M, splits, a["n_frames"], b["n_frames"])
        pg.append(a["pcm"]); po.append(b["pcm"]); co.append(b["chan"]); pos += s
        if tap:
            fg.append(a["fm"]); fo.append(b["fm"])
        seen += a["n_frames"]
        if "rssi" in want and a["n_frames"] > 8:
            act = [k for k in (ks or range(M)) if synth.channel_kind(k) != "empty"]
            assert np.abs(a["rssi"][act] - b["rssi"][act]).max() < 0.05, (fs, M, "rssi")
    pg, po, co = np.concatenate(pg, axis=1), np.concatenate(po, axis=1), np.concatenate(co, axis=1)
    fg, fo = np.concatenate(fg, axis=1), np.concatenate(fo, axis=1)
    act = [k for k in (ks or range(M)) if synth.channel_kind(k) != "empty"]
    d, reached = 0, 0
    if pg.shape[1]:
        h = parity_rule.audio_response(HP, 4.0, B0, B1, A1, deemph_fir_taps=DE_FIR if opts.get("deemph_fir") else None,
                                       lp_taps=LP if opts.get("lowpass") else None)
        v = parity_rule.check(pg[act], po[act], co[act], fg[act], fo[act], h)
        if "error" in v:
            print("FAIL (rule) %r" % v["error"], flush=True); sys.exit(1)
        d = v["max_abs_pcm_diff_lsb"] if v["ok"] else max(2, v["max_abs_pcm_diff_lsb"], int(np.ceil(v["ill_conditioned"]["max_abs_unexplained_lsb"])))
        reached = v["ill_conditioned"]["samples_over_1_lsb"]
    worst = max(worst, d)
    n_cases += 1
    status = "ok" if d <= 1 else "FAIL"
    print("%s fs=%g M=%d opts=%s max_block=%d splits=%s frames=%d maxdiff=%d%s" % (status, fs, M, opts, max_block, splits, pg.shape[1], d,
          " (%d samples > 1 LSB explained by ill-conditioned start-up discriminator samples)" % reached if reached else ""), flush=True)
    g.close(); o.close()
    if d > 1:
        sys.exit(1)
print("soak: %d cases, worst |pcm diff| = %d LSB" % (n_cases, worst))
