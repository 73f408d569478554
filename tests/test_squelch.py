"""the product's squelch / channel-select host logic (C-ABI) against the oracle restatement of
src/sdr_pmr446.c:668-700, :828-874 on random RSSI sequences, incl. masks, lock modes and the 18/13 dB hysteresis."""
import ctypes as C

import numpy as np
import pytest

from oracle import squelch as orc
from sdr_pmr446_amd import chain


def _words(mask, M):
    """Python int (bit k = channel k) -> the uint64 mask words of pmr_chain_set_channel_mask / pmr_squelch_update."""
    n = (M + 63) // 64
    return np.array([(mask >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(n)], dtype=np.uint64)


@pytest.mark.parametrize("M,lock_max,mask", [(16, 0, 0xFFFF), (16, 1, 0xFFFF), (16, 1, 0x0FF3), (256, 0, (1 << 256) - 1 - 0xF),
                                             (256, 1, ((1 << 256) - 1) & ~(0xFF << 64) & ~(1 << 200)),     # channels >= 64 CAN be excluded
                                             (1024, 0, sum(1 << k for k in range(0, 1024, 3)))])
def test_squelch_state_machine_matches_reference_logic(M, lock_max, mask):
    L = chain.load()
    mw = _words(mask, M)
    rng = np.random.default_rng(M + lock_max)
    s = chain.Squelch()
    L.pmr_squelch_init(C.byref(s))
    o = orc.Squelch()
    assert (s.state, s.active_chan) == (0, -1)
    seen_tuned = seen_detune = False
    for step in range(400):
        rssi = (rng.standard_normal(M) * 2.0 - 30.0).astype(np.float32)
        if (step // 25) % 2 == 1:                       # a carrier comes and goes, sometimes hopping
            rssi[(3 + 67 * (step // 50)) % M] += np.float32(rng.uniform(10.0, 40.0))     # (lands on masked-out channels too)
        changed = L.pmr_squelch_update(C.byref(s), rssi.ctypes.data, M, mw.ctypes.data, len(mw), 18.0, lock_max)
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
    m = np.array([0b1011], dtype=np.uint64)
    assert L.pmr_find_max_rssi_channel(rssi.ctypes.data, 4, m.ctypes.data, 1, C.byref(mr)) == 0      # channel 2 masked out
    assert mr.value == pytest.approx(-10 - (-10 - 50 - 50) / 3.0, abs=1e-5)
    z = np.zeros(1, dtype=np.uint64)
    assert L.pmr_find_max_rssi_channel(rssi.ctypes.data, 4, z.ctypes.data, 1, C.byref(mr)) == -1
    assert L.pmr_find_max_rssi_channel(rssi.ctypes.data, 4, None, 0, C.byref(mr)) == 2                # NULL = every channel
    # a channel beyond 64 excluded (round 3 treated every channel >= 64 as enabled), and a mask shorter than M refused
    big = np.full(256, -40.0, dtype=np.float32); big[200] = 0.0; big[70] = -5.0
    w = _words(((1 << 256) - 1) & ~(1 << 200), 256)
    assert L.pmr_find_max_rssi_channel(big.ctypes.data, 256, w.ctypes.data, 4, C.byref(mr)) == 70
    assert L.pmr_find_max_rssi_channel(big.ctypes.data, 256, w.ctypes.data, 3, C.byref(mr)) == -1


def test_channel_selection_is_the_references_arithmetic_on_random_masks():
    """pmr_find_max_rssi_channel walks the mask word by word (its own structure: lowest open channel first by ctz); the numbers must be the
    reference's, bit for bit -- src/sdr_pmr446.c:668-700: float32 sum in channel order, strict '>' (ties to the lowest channel), max - mean --
    for any M, multi-word masks, masks with closed words, ties, and no open channel at all (margin untouched)."""
    import ctypes as C
    import numpy as np
    from sdr_pmr446_amd import chain
    L = chain.load()
    L.pmr_find_max_rssi_channel.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_float)]
    L.pmr_find_max_rssi_channel.restype = C.c_int
    rng = np.random.default_rng(1)

    def reference(r, M, mask):
        best, mx, acc, n = -1, np.float32(0), np.float32(0), 0
        for i in range(M):
            if mask is not None and not (int(mask[i >> 6]) >> (i & 63)) & 1:
                continue
            n += 1
            acc = np.float32(acc + r[i])
            if best < 0 or r[i] > mx:
                mx, best = r[i], i
        return (best, np.float32(mx - np.float32(acc / np.float32(n)))) if best >= 0 else (-1, None)

    for _ in range(600):
        M = int(rng.choice([1, 4, 16, 63, 64, 65, 100, 128, 256, 1000, 1024, 4096]))
        r = (rng.standard_normal(M) * 10 - 50).astype(np.float32)
        if rng.random() < 0.3:
            r[rng.integers(M)] = r.max()
        nw = (M + 63) // 64
        mask = None if rng.random() < 0.2 else (rng.integers(0, 2 ** 63, size=nw, dtype=np.uint64) * np.uint64(2)
                                                 + rng.integers(0, 2, size=nw).astype(np.uint64))
        if mask is not None and rng.random() < 0.2:
            mask[:] = 0
        out = C.c_float(123.0)
        got = L.pmr_find_max_rssi_channel(r.ctypes.data, M, mask.ctypes.data if mask is not None else None, nw, C.byref(out))
        want, margin = reference(r, M, mask)
        assert got == want
        assert (out.value == 123.0) if want < 0 else (np.float32(out.value) == margin)
