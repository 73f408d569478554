"""Device selection at the boundary (include/pmr_chain.h: pmr_chain_cfg.device = ordinal, or -1 = the calling thread's current
device): one handle = one IQ stream = one GPU is the multi-GPU model (the reference's loop per stream,
src/sdr_pmr446.c:788-908), so what a wrong or implicit ordinal does must be defined: NULL + pmr_chain_create_error(), never a crash,
never another device silently."""
import ctypes as C

import numpy as np
import pytest

from parity_util import CFG2
from sdr_pmr446_amd import synth

pytestmark = pytest.mark.gpu


def test_device_minus_one_is_the_current_device_and_equals_the_explicit_ordinal():
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n = 200000
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0)
    probe = chain.DeviceBuffer(256, 0)                         # pmr_device_alloc(.., 0): hipSetDevice(0) in the library's runtime
    outs = []
    for dev in (-1, 0):
        g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n, device=dev)
        outs.append(g.process_block(x, want=("pcm", "rssi")))
        g.close()
    probe.free()
    assert np.array_equal(outs[0]["pcm"], outs[1]["pcm"]) and np.array_equal(outs[0]["rssi"], outs[1]["rssi"])
    assert np.abs(outs[0]["pcm"]).max() > 1000


@pytest.mark.parametrize("dev,what", [(4096, "does not exist"), (-2, "-1 (the calling thread's current device)")])
def test_a_device_that_does_not_exist_is_refused_with_a_reason(dev, what):
    from sdr_pmr446_amd import chain
    L = chain.load()
    cfg = chain.make_cfg(fs_in=CFG2[0], num_channels=CFG2[1], max_block=100000, device=dev)
    assert not L.pmr_chain_create(C.byref(cfg))                # NULL, no crash
    msg = L.pmr_chain_create_error().decode()
    assert what in msg and str(dev) in msg, msg
    with pytest.raises(chain.PmrError, match="device ordinal"):
        chain.PmrChain(fs_in=CFG2[0], num_channels=CFG2[1], device=dev)
    # ... and the failure leaves nothing behind: the next create works, the reason is cleared
    g = chain.PmrChain(fs_in=CFG2[0], num_channels=CFG2[1], max_block=100000, device=0)
    assert L.pmr_chain_create_error() == b""
    assert g.process_block(synth.synth_iq(100000, *CFG2), want=("pcm",))["n_frames"] > 500
    g.close()
    assert not chain.load().pmr_device_alloc(256, dev)         # include/pmr_mem.h: NULL on a bad ordinal too
