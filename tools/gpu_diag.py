#!/usr/bin/env python3
"""Stage-by-stage parity report: HIP chain vs CPU oracle on identical synthetic IQ (run on the GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from parity_util import active_channels, pcm_diff, rel_err, run_blocks  # noqa: E402
from sdr_pmr446_amd import chain, synth  # noqa: E402

WANT = ("pcm", "audio", "chan", "rssi", "resampled", "fm")


def report(tag, fs, M, N, splits, dev_hz=2500.0, synth_ch=None, **kw):
    x = synth.synth_iq(N, fs, M, channels=synth_ch, dev_hz=dev_hz)
    mb = max(splits)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb, **kw)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb, **kw)
    t = time.time(); ro = run_blocks(o, x, splits, WANT); to = time.time() - t
    t = time.time(); rg = run_blocks(g, x, splits, WANT); tg = time.time() - t
    act = active_channels(M, synth_ch)
    print("[%s] fs=%.4g M=%d N=%d blocks=%d frames o=%d g=%d  (oracle %.2fs, hip %.2fs)" %
          (tag, fs, M, N, len(splits), ro["n_frames"], rg["n_frames"], to, tg))
    ok = ro["n_frames"] == rg["n_frames"] and len(ro["resampled"]) == len(rg["resampled"])
    if not ok:
        print("   COUNT MISMATCH resampled o=%d g=%d" % (len(ro["resampled"]), len(rg["resampled"])))
        return False
    e_res = rel_err(rg["resampled"], ro["resampled"])
    e_ch = rel_err(rg["chan"], ro["chan"])
    e_fm = float(np.abs(rg["fm"][act] - ro["fm"][act]).max()) if ro["n_frames"] else 0.0
    e_au = float(np.abs(rg["audio"][act] - ro["audio"][act]).max()) if ro["n_frames"] else 0.0
    d = pcm_diff(rg["pcm"][act], ro["pcm"][act])
    e_rssi = max(float(np.abs(a - b)[act].max()) for a, b in zip(rg["rssi"], ro["rssi"])) if ro["n_frames"] else 0.0
    sat = float((np.abs(ro["pcm"][act]) >= 32767).mean()) if d.size else 0.0
    print("   resampled rel %.3g | chan rel %.3g | fm abs %.3g | audio abs %.3g | rssi dB %.3g | pcm maxdiff %d (frac!=0 %.4f, saturated %.3f)"
          % (e_res, e_ch, e_fm, e_au, e_rssi, int(d.max()) if d.size else 0, float((d > 0).mean()) if d.size else 0, sat))
    if ro["n_frames"]:
        dfm = np.abs(rg["fm"][act] - ro["fm"][act]); ij = np.unravel_index(dfm.argmax(), dfm.shape)
        dau = np.abs(rg["audio"][act] - ro["audio"][act]); ia = np.unravel_index(dau.argmax(), dau.shape)
        print("   fm max err at ch %d frame %d (|chan|=%.3g) ; audio max err at ch %d frame %d ; fm err after frame 40: %.3g ; audio err after frame 450: %.3g"
              % (act[ij[0]], ij[1], abs(ro["chan"][act[ij[0]], ij[1]]), act[ia[0]], ia[1],
                 dfm[:, 40:].max() if dfm.shape[1] > 40 else 0, dau[:, 450:].max() if dau.shape[1] > 450 else 0))
    good = e_res < 2e-5 and e_ch < 2e-5 and e_fm < 1e-4 and (d.size == 0 or d.max() <= 1)
    print("   ->", "OK" if good else "FAIL")
    o.close(); g.close()
    return good


def main():
    rng = np.random.default_rng(1)
    allok = True
    allok &= report("cfg2 one block", 2.4e6, 16, 100000, [100000], dev_hz=500.0)
    allok &= report("cfg2 default dev (saturating)", 2.4e6, 16, 100000, [100000])
    sp = []
    left = 300000
    while left:
        n = int(min(left, rng.integers(1, 60000)))
        sp.append(n); left -= n
    allok &= report("cfg2 random splits", 2.4e6, 16, 300000, sp, dev_hz=500.0)
    allok &= report("cfg2 tiny blocks", 2.4e6, 16, 20000, [1, 7, 0, 100, 4095, 4096, 4097, 3000, 16, 8, 4580], dev_hz=500.0)
    allok &= report("ref point", 1.024e6, 16, 200000, [100000, 100000], dev_hz=500.0)
    allok &= report("cfg2 lowpass", 2.4e6, 16, 200000, [100000, 100000], dev_hz=500.0, lowpass=True)
    allok &= report("cfg2 fir deemph+lp", 2.4e6, 16, 200000, [100000, 100000], dev_hz=500.0, lowpass=True, deemph_fir=True)
    allok &= report("cfg3", 61.44e6, 256, 1 << 22, [1 << 21, 1 << 21], dev_hz=500.0, synth_ch=list(range(0, 256, 5)))
    allok &= report("cfg5", 1.0e9, 1024, 1 << 25, [1 << 24, 1 << 24], dev_hz=500.0, synth_ch=list(range(0, 1024, 73)))
    print("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
