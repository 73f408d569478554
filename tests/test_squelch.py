"""the product's squelch / channel-select host logic (C-ABI) against the oracle restatement of
src/sdr_pmr446.c:668-700, :828-874 on random RSSI sequences, incl. masks, lock modes and the 18/13 dB hysteresis."""
import ctypes as C

import numpy as np
import pytest

from oracle import squelch as orc
from sdr_pmr446_amd import chain


@pytest.mark.parametrize("M,lock_max,mask", [(16, 0, 0xFFFF), (16, 1, 0xFFFF), (16, 1, 0x0FF3), (256, 0, 0xFFFFFFFFFFFFFFF0)])
def test_squelch_state_machine_matches_reference_logic(M, lock_max, mask):
    L = chain.load()
    rng = np.random.default_rng(M + lock_max)
    s = chain.Squelch()
    L.pmr_squelch_init(C.byref(s))
    o = orc.Squelch()
    assert (s.state, s.active_chan) == (0, -1)
    seen_tuned = seen_detune = False
    for step in range(400):
        rssi = (rng.standard_normal(M) * 2.0 - 30.0).astype(np.float32)
        if (step // 25) % 2 == 1:                       # a carrier comes and goes, sometimes hopping
            rssi[(3 + step // 50) % M] += np.float32(rng.uniform(10.0, 40.0))
        changed = L.pmr_squelch_update(C.byref(s), rssi.ctypes.data, M, mask, 18.0, lock_max)
        ochanged = o.update(rssi, mask, 18.0, lock_max)
        assert (s.state, s.active_chan, bool(changed)) == (o.state, o.active_chan, bool(ochanged)), step
        assert s.rssi == pytest.approx(float(o.rssi), abs=1e-4)
        seen_tuned |= s.state == 1
        seen_detune |= bool(changed) and s.active_chan == -1
    assert seen_tuned and seen_detune


def test_find_max_respects_mask_and_all_disabled():
    L = chain.load()
    rssi = np.array([-10, -50, 5, -50], dtype=np.float32)
    mr = C.c_float(123.0)
    assert L.pmr_find_max_rssi_channel(rssi.ctypes.data, 4, 0b1011, C.byref(mr)) == 0      # channel 2 masked out
    assert mr.value == pytest.approx(-10 - (-10 - 50 - 50) / 3.0, abs=1e-5)
    assert L.pmr_find_max_rssi_channel(rssi.ctypes.data, 4, 0, C.byref(mr)) == -1
