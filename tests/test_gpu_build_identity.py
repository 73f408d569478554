"""Round 6's front-end changes -- half-band stages computed straight from registers (hb_stage_reg: DPP halo instead of an LDS round trip)
and level 1's 32 KB tile layout (scratch inside each wave's own quarter of the dead raw tile: five tiles per CU) -- keep every
operation and its order, so they must not change a single output bit.  This test BUILDS the round-5 form of the kernels from the same
sources (-DFE_S1_LDS -DFE_L1_LDS23 -DFE_NO_TIGHT: the stages through LDS, 33.7 KB tiles; an experiment build under build_ab/) and
compares sha256 of the PCM of three ragged un-synchronised calls and of the resampled stream, on the reference's operating point
(m = 5, 10 only), cfg2, cfg3 and cfg5 (tools/pcm_hash.py).  Reference stages: dc-block + msresamp_crcf, src/sdr_pmr446.c:795-796."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.nopoison]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hashes(lib=None):
    env = {k: v for k, v in os.environ.items() if k not in ("PMR_LIBRARY", "PMR_DEBUG_POISON")}
    if lib:
        env["PMR_LIBRARY"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pcm_hash.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    out = {l.split()[0]: l for l in r.stdout.splitlines() if l.strip()}
    assert set(out) == {"ref", "cfg2", "cfg3", "cfg5"}, r.stdout
    return out


def test_register_stages_and_the_32k_tile_change_no_bit():
    from sdr_pmr446_amd import build
    lib = build.build_variant("r5form", "-DFE_S1_LDS -DFE_L1_LDS23 -DFE_NO_TIGHT")
    new, old = _hashes(), _hashes(lib)
    assert new == old, (new, old)
