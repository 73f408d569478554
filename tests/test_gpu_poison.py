"""The poison mode of the -m gpu tier checks itself (csrc/pmr_poison.hip, tests/conftest.py::_gpu_poison).

Round 3's red driver run was `k_ct_goertzel` reading LDS beyond its workgroup's allocation and multiplying what it found by zero
(reference detector: src/sdr_pmr446.c:366-409): green wherever the stale bytes were ordinary floats, red where they were a NaN.
Under the poison mode every such read finds a NaN, on every box.  "The suite is green under poison" only means something if the
poison really lands, so: (1) a probe kernel that copies its UNINITIALISED LDS out must read the pattern from every workgroup on
every CU; (2) scratch allocations are 0xFF-filled; (3) the mode changes no result."""
import os

import numpy as np
import pytest

from parity_util import CFG2, CFG3

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("PMR_TEST_POISON", "1") == "0", reason="the poison mode was switched off for this run")]

POISON = 0x7FA0DEAD


def test_every_workgroup_on_every_cu_starts_on_poisoned_lds():
    from sdr_pmr446_amd import chain
    L = chain.load()
    assert L.pmr_debug_poison(1) == 1, "conftest switches the mode on for every -m gpu test"
    # 32 KB x 5 per CU, 64 KB x 2 per CU, 8 KB x 20 per CU, and the WHOLE 160 KB of every CU (one workgroup each, two rounds): a poison
    # kernel that was granted less than 160 KB per workgroup would leave the top of every CU's LDS as it was
    for words, n_wg in [(8192, 1024), (16384, 512), (2048, 4096), (40960, 512)]:
        out = chain.DeviceBuffer(words * n_wg * 4)
        assert L.pmr_debug_lds_probe(out.ptr, words, n_wg) == 0
        got = out.download(np.uint32, words * n_wg)
        out.free()
        bad = np.flatnonzero(got != POISON)
        assert bad.size == 0, "%d of %d LDS words were not poisoned (first: wg %d word %d = %#x)" % (
            bad.size, got.size, bad[0] // words, bad[0] % words, got[bad[0]])
    assert np.isnan(np.array([POISON], dtype=np.uint32).view(np.float32)[0])


def test_scratch_allocations_are_filled_with_ff():
    from sdr_pmr446_amd import chain
    buf = chain.DeviceBuffer(4096)
    got = buf.download(np.uint8, 4096)
    buf.free()
    assert (got == 0xFF).all()


@pytest.mark.parametrize("cfg,n", [(CFG2, 400000), (CFG3, 1 << 21)], ids=["cfg2", "cfg3"])
def test_poison_mode_changes_no_result(cfg, n):
    from sdr_pmr446_amd import chain, synth
    fs, M = cfg
    L = chain.load()
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0, channels=list(range(0, M, max(1, M // 16))))
    outs = []
    for on in (1, 0):
        L.pmr_debug_poison(on)
        g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
        r = g.process_block(x, want=("pcm", "rssi", "ctcss"))
        ev = r["ctcss"]
        g.close()
        outs.append((r["pcm"].copy(), np.asarray(r["rssi"]).copy(), ev.copy()))
    L.pmr_debug_poison(1)
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2].tobytes() == outs[1][2].tobytes()
    assert np.isfinite(outs[0][2]["max_power"]).all() and np.isfinite(outs[0][2]["avg_power"]).all()
