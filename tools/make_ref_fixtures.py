#!/usr/bin/env python3
"""tools/make_ref_fixtures.py -- golden vectors from the REFERENCE'S OWN CODE, for the parts of the path that are not liquid-dsp.

Runs in the BUILD CONTAINER only (it reads /root/reference; nothing of the reference travels to the GPU box -- the fixtures
are data).  Two pieces of the hot path's neighbourhood are plain C / Python in the reference and can be executed here:

 (i)  the CTCSS tone detector: ctcss_detector_reset / _create / _analyze, src/sdr_pmr446.c:338-409, with its
      struct include/sdr_pmr446.h:42-52, tone table :138-141 and constants :14,:24,:37,:46.  Those LINE RANGES are cut out of the
      reference at run time into a temporary file, compiled with gcc next to a 20-line driver of ours (stdin floats -> one line
      per Goertzel block), and fed the detector's input of a synthetic tone-level sweep: the oracle chain's low-pass branch
      (`ctcss_lp`, :889) through the oracle's dc blocker (:606).  Output: tests/golden/ctcss_ref.npz --
        x          [K][N] float32   detector input of K channels (the oracle's restatement is checked on exactly these samples)
        x_channels [K]
        index / detected / max_power [16][B], power [16][B][38]   what the REFERENCE's code decided, per channel and block
        synth      the parameters that regenerate the IQ (the GPU test feeds the same signal to the HIP chain)
      The binary is kept as oracle/_ref/ctcss_ref (git-ignored).
 (ii) the de-emphasis coefficients (row a8): scripts/filter_des.py:31-44 standard_deemph() is imported (matplotlib on the Agg
      backend, its plots and prints swallowed) and evaluated: tests/golden/deemph_ref.npz {b, a, tau, fs}.

 (iii) average_power (row f1): src/sdr_pmr446.c:330-336 cut out and compiled the same way, run on the channelizer output rows of a
      synthetic block: tests/golden/rssi_ref.npz {chan, rssi_db}.

    python3 tools/make_ref_fixtures.py            regenerate the fixtures
    python3 tools/make_ref_fixtures.py --build    only compile oracle/_ref/ctcss_ref (what __graft_entry__.build() calls)
"""
import contextlib
import io
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_C = os.path.join(REF, "src", "sdr_pmr446.c")
REF_H = os.path.join(REF, "include", "sdr_pmr446.h")
OUT_BIN = os.path.join(ROOT, "oracle", "_ref", "ctcss_ref")

DRIVER = r"""
/* driver (ours): float32 samples on stdin -> one text line per completed Goertzel block */
#include <stdio.h>
int main(void)
{
    ctcss_detector_t *d = ctcss_detector_create();
    float x;
    if (!d) return 1;
    while (fread(&x, sizeof(x), 1, stdin) == 1) {
        ctcss_detector_analyze(d, &x, 1);
        if (d->samp_processed == 0) {
            printf("%d %d %.9g", d->max_power_index, d->tone_detected ? 1 : 0, (double)d->max_power);
            for (int j = 0; j < CTCSS_NUM_FREQS; ++j) printf(" %.9g", (double)d->power[j]);
            printf("\n");
        }
    }
    free(d);
    return 0;
}
"""


def _lines(path, lo, hi):
    with open(path) as f:
        return "".join(f.readlines()[lo - 1:hi])


def _define(path, name):
    with open(path) as f:
        for l in f:
            if l.startswith("#define " + name + " ") or l.startswith("#define " + name + "\t"):
                return l
    raise SystemExit("no #define %s in %s" % (name, path))


def build_ctcss_ref():
    """gcc the reference's detector (line ranges cut out at run time) + our driver -> oracle/_ref/ctcss_ref"""
    os.makedirs(os.path.dirname(OUT_BIN), exist_ok=True)
    struct = _lines(REF_H, 42, 52)
    assert "ctcss_detector_t" in struct and "samp_processed" in struct, "include/sdr_pmr446.h:42-52 is not the detector struct"
    freqs = _lines(REF_C, 138, 141)
    assert "ctcss_freqs" in freqs and "250.3" in freqs
    body = _lines(REF_C, 338, 409)
    assert body.lstrip().startswith("static void ctcss_detector_reset") and "tone_detected =" in body
    # the constants the cut-out uses, as the reference defines them (:14 of the header; :22-24, :37, :46 of the source)
    defs = _define(REF_H, "CTCSS_NUM_FREQS") + "".join(_define(REF_C, n) for n in (
        "CHANNEL_WIDTH_HZ", "AUDIO_SAMPLERATE", "SDR_CHANNEL_BUF_SIZE", "CTCSS_BLOCK_SIZE"))
    src = ("#include <math.h>\n#include <stdbool.h>\n#include <stddef.h>\n#include <stdlib.h>\n"
           + defs + struct + freqs + body + DRIVER)
    with tempfile.TemporaryDirectory() as td:                  # the cut-out never lands in the repository
        c = os.path.join(td, "ctcss_ref.c")
        with open(c, "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-o", OUT_BIN, c, "-lm"])
    return OUT_BIN


def run_ctcss_ref(x):
    """the reference's decisions for one float32 stream: (index[B], detected[B], max_power[B], power[B][38])"""
    out = subprocess.run([OUT_BIN], input=np.ascontiguousarray(x, dtype=np.float32).tobytes(), capture_output=True, check=True).stdout
    rows = [l.split() for l in out.decode().splitlines()]
    idx = np.array([int(r[0]) for r in rows], dtype=np.int32)
    det = np.array([int(r[1]) for r in rows], dtype=np.int32)
    mx = np.array([float(r[2]) for r in rows], dtype=np.float32)
    pw = np.array([[float(v) for v in r[3:]] for r in rows], dtype=np.float32).reshape(len(rows), 38)
    return idx, det, mx, pw


SYNTH = dict(fs=2.4e6, M=16, n=2400000, dev_hz=1500.0,
             ctcss_devs=[150, 200, 230, 245, 250, 252, 255, 260, 280, 300, 320, 340, 400, 500, 600, 700])
X_CHANNELS = [0, 5, 9, 13]            # detector inputs stored in full: below, at, above the avg-power threshold, far above


def detector_inputs():
    """[16][N] float32: what ctcss_detector_analyze sees on every channel of the synthetic sweep (oracle chain: low-pass branch
    :884-889, then the dc blocker of ctcss_execute :606)"""
    sys.path.insert(0, ROOT)
    import oracle
    from sdr_pmr446_amd import synth
    p = SYNTH
    x = synth.synth_iq(p["n"], p["fs"], p["M"], dev_hz=p["dev_hz"], ctcss_dev_of=lambda k: p["ctcss_devs"][k])
    o = oracle.OracleChain(fs_in=p["fs"], num_channels=p["M"], max_block=p["n"])
    lp = o.process_block(x, want=("pcm", "ctcss_lp"))["ctcss_lp"]
    o.close()
    L = oracle.lib()
    out = np.empty_like(lp)
    import ctypes as C
    L.orc_dcblock_rrrf_run.argtypes = [C.c_void_p, C.c_uint, C.c_float, C.c_void_p]
    L.orc_dcblock_rrrf_run.restype = None
    for k in range(lp.shape[0]):
        row = np.ascontiguousarray(lp[k])
        L.orc_dcblock_rrrf_run(row.ctypes.data, len(row), 0.0005, out[k].ctypes.data)
    return out


def make_ctcss_fixture():
    build_ctcss_ref()
    xin = detector_inputs()
    M, N = xin.shape
    B = N // 2441
    res = [run_ctcss_ref(xin[k]) for k in range(M)]
    assert all(len(r[0]) == B for r in res) and B >= 3
    path = os.path.join(ROOT, "tests", "golden", "ctcss_ref.npz")
    nkeep = B * 2441
    np.savez_compressed(
        path, x=xin[X_CHANNELS, :nkeep].astype(np.float32), x_channels=np.array(X_CHANNELS, dtype=np.int32),
        index=np.stack([r[0] for r in res]), detected=np.stack([r[1] for r in res]), max_power=np.stack([r[2] for r in res]),
        power=np.stack([r[3] for r in res]),
        synth_fs=SYNTH["fs"], synth_M=SYNTH["M"], synth_n=SYNTH["n"], synth_dev_hz=SYNTH["dev_hz"],
        synth_ctcss_devs=np.array(SYNTH["ctcss_devs"], dtype=np.float64),
        source=np.array("reference src/sdr_pmr446.c:338-409 + include/sdr_pmr446.h:42-52 compiled by tools/make_ref_fixtures.py"))
    det = np.stack([r[1] for r in res])
    print("ctcss_ref.npz: %d channels x %d blocks, detected %d / not %d, %d bytes" %
          (M, B, int(det[:, 1:].sum()), int((1 - det[:, 1:]).sum()), os.path.getsize(path)))


RSSI_DRIVER = r"""
/* driver (ours): [uint32 rows][uint32 len] then rows x len complex float32 on stdin -> one average_power() per row */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(void)
{
    uint32_t hdr[2];
    if (fread(hdr, 4, 2, stdin) != 2) return 1;
    complex float *row = malloc((size_t)hdr[1] * sizeof(*row));
    for (uint32_t r = 0; r < hdr[0]; ++r) {
        if (fread(row, sizeof(*row), hdr[1], stdin) != hdr[1]) return 2;
        printf("%.9g\n", (double)average_power(row, hdr[1]));
    }
    return 0;
}
"""
OUT_RSSI = os.path.join(ROOT, "oracle", "_ref", "rssi_ref")


def build_rssi_ref():
    """gcc the reference's average_power (src/sdr_pmr446.c:330-336, cut out at run time) + our driver -> oracle/_ref/rssi_ref"""
    os.makedirs(os.path.dirname(OUT_RSSI), exist_ok=True)
    body = _lines(REF_C, 330, 336)
    assert body.lstrip().startswith("static float average_power") and "log10f" in body
    src = "#include <complex.h>\n#include <math.h>\n#include <stddef.h>\n" + body + RSSI_DRIVER
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "rssi_ref.c")
        with open(c, "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-o", OUT_RSSI, c, "-lm"])
    return OUT_RSSI


def make_rssi_fixture():
    """(iii) average_power (row f1, src/sdr_pmr446.c:330-336) on the channelizer output rows of a synthetic block: the numbers the
    reference's find_max_rssi_channel (:668-700) compares.  tests/golden/rssi_ref.npz: chan [M][ns] complex64 (the oracle chain's
    tap-off), rssi_db [M] as the REFERENCE's code computes them, synth parameters (the GPU test regenerates the IQ)."""
    build_rssi_ref()
    sys.path.insert(0, ROOT)
    import oracle
    from sdr_pmr446_amd import synth
    fs, M, n = 2.4e6, 16, 100000
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n)
    r = o.process_block(x, want=("pcm", "chan", "rssi"))
    o.close()
    chan = np.ascontiguousarray(r["chan"], dtype=np.complex64)
    hdr = np.array([chan.shape[0], chan.shape[1]], dtype=np.uint32).tobytes()
    out = subprocess.run([OUT_RSSI], input=hdr + chan.tobytes(), capture_output=True, check=True).stdout
    rssi = np.array([float(l) for l in out.decode().split()], dtype=np.float32)
    assert rssi.shape == (M,)
    path = os.path.join(ROOT, "tests", "golden", "rssi_ref.npz")
    np.savez_compressed(path, chan=chan, rssi_db=rssi, oracle_rssi_db=np.asarray(r["rssi"], dtype=np.float32),
                        synth_fs=fs, synth_M=M, synth_n=n, synth_dev_hz=1500.0,
                        source=np.array("reference src/sdr_pmr446.c:330-336 average_power() compiled by tools/make_ref_fixtures.py"))
    print("rssi_ref.npz: %d channels x %d frames, rssi %.2f .. %.2f dB, max |oracle - reference| = %.3g dB, %d bytes" %
          (M, chan.shape[1], rssi.min(), rssi.max(), float(np.abs(rssi - r["rssi"]).max()), os.path.getsize(path)))


def make_deemph_fixture():
    import importlib.util
    os.environ["MPLBACKEND"] = "Agg"
    spec = importlib.util.spec_from_file_location("ref_filter_des", os.path.join(REF, "scripts", "filter_des.py"))
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):            # the script prints and plots at import time
        spec.loader.exec_module(mod)
    b, a = mod.standard_deemph()
    path = os.path.join(ROOT, "tests", "golden", "deemph_ref.npz")
    np.savez(path, b=np.array(b, dtype=np.float64), a=np.array(a, dtype=np.float64), tau=50e-6, fs=float(mod.FS),
             source=np.array("reference scripts/filter_des.py:31-44 standard_deemph(), imported by tools/make_ref_fixtures.py"))
    print("deemph_ref.npz: b =", list(b), "a =", list(a))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("this tool runs in the build container (needs %s)" % REF)
    if "--build" in sys.argv:
        print(build_ctcss_ref())
        print(build_rssi_ref())
    else:
        make_ctcss_fixture()
        make_deemph_fixture()
        make_rssi_fixture()
