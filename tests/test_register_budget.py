"""The kernels' VGPR counts are part of the design: a SIMD has 512 registers, allocated in granules of 8.  Four front-end tiles at <= 96
(= 384) leave 128 for one back-end wave, so the front end must stay <= 96 and every back-end kernel of the hot path <= 128 -- a build that
crossed either line by five registers cost the cfg2 chain 8-10 % with no test failing (profiles/r03_ab_log.txt).  (Rounds
3-5 held the front end to 88 and the back end to 160; round 6's stage 1 from registers needs 89 = the same 96-register granule, and no
back-end kernel of the chain is above 128: profiles/r06_ab_log.txt r6e.)  Cross-compiles the units for
gfx950 (no GPU needed) and reads the counts from the code objects' metadata."""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdr_pmr446_amd", "csrc")
HIPCC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")

BUDGET = [                      # (unit, regex on the mangled name, max VGPRs, why)
    ("pmr_fe_fast.hip", r"k_fe_fastILi0ELi[1-9]E", 96, "one-level front end: four tiles per SIMD (4 x 96) + one 128-register back-end wave"),
    ("pmr_fe_fast.hip", r"k_fe_fastILi0ELi0E", 96, "one-level front end of the reference's own 1.024 MS/s plan (small blocks: its back end is the 64- / 84-register small-block kernels)"),
    ("pmr_fe_fast.hip", r"k_fe_fastILi1E", 64, "level 1 of the two-level front end"),
    ("pmr_fe_fast.hip", r"k_fe_level2ILi5ELi10E", 80, "level 2 (the reference's As = 60 pair) runs beside four level-1 tiles"),
    ("pmr_fe_fast.hip", r"k_fe_level2", 96, "level 2 for the other (MA, MB) pairs: five waves per SIMD, still beside four 54-register level-1 tiles"),
    ("pmr_fir_fft.hip", r"k_fir_fftILi4ELb0E", 128, "FFT form of the audio FIR (1024 points): one-wave workgroups, four per SIMD, beside four front-end tiles"),
    ("pmr_fir_fft.hip", r"k_fir_fftILi8ELb0E", 128, "FFT form of the audio FIR (2048 points): two-wave workgroups beside four front-end tiles"),
    ("pmr_fir_mfma4.hip", r"k_fir_mfma4ILb0ELb0ELb0E", 64, "128-frame audio FIR: four workgroups per CU"),
    ("pmr_channelize_wide.hip", r"k_pfb_wide", 128, "1024-channel bank: beside four level-1 tiles"),
    ("pmr_channelize_wide.hip", r"k_fft_disc", 128, "FFT + discriminator of the wide banks"),
    ("pmr_channelize_small.hip", r"k_channelize_winILi16ELi26ELb1ELi16E", 88, "16-channel bank: a wave fits beside four front-end tiles with room to spare"),
    ("pmr_channelize_wide.hip", r"k_channelize_fused256ILb1E", 128, "256-channel bank: four waves per SIMD"),
]


def _counts(unit):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        flags = ["-fno-slp-vectorize"]                      # the product's flags (build.PRODUCT_HIP_FLAGS)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S"] + flags +
                              [os.path.join(CSRC, unit), "-o", out], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    res = {}
    for m in re.finditer(r"\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        res[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_kernels_stay_inside_their_register_budgets():
    units = sorted({b[0] for b in BUDGET})
    with ThreadPoolExecutor(max_workers=4) as ex:
        counts = dict(zip(units, ex.map(_counts, units)))
    for unit, pat, limit, why in BUDGET:
        hits = {k: v for k, v in counts[unit].items() if re.search(pat, k)}
        assert hits, (unit, pat)
        for name, (vgpr, spill) in hits.items():
            assert spill == 0, (name, "spills")
            assert vgpr <= limit, "%s uses %d VGPRs, budget %d (%s)" % (name, vgpr, limit, why)
