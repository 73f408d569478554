"""SURVEY s8 row f3 on the GPU: the `dsd_in` loop body (reference src/dsd_in.c:167-175) through the C-ABI of
include/pmr_dsd.h against the oracle restatement -- int16 PCM within +-1 LSB on identical synthetic IQ, ragged block
splits, other input / output rates (deeper cascade, two half-band interpolators, none), and the committed golden."""
import numpy as np
import pytest

import oracle
from parity_util import pcm_diff, rel_err
from sdr_pmr446_amd import synth
from test_dsd_cpu import load_golden

pytestmark = pytest.mark.gpu


def run(obj, x, splits, want=("pcm", "audio")):
    outs, pos = {}, 0
    for n in splits:
        o = obj.process_block(x[pos:pos + n], want=want)
        pos += n
        for k, v in o.items():
            outs.setdefault(k, []).append(v)
    assert pos == len(x)
    return {k: (np.concatenate(v) if k not in ("n_out", "n_resampled") else v) for k, v in outs.items()}


@pytest.mark.parametrize("fs,audio_rate,n,splits", [
    (1.024e6, 48000.0, 600000, [200000, 200000, 200000]),                       # the reference's operating point
    (1.024e6, 48000.0, 450001, [1, 199999, 0, 63, 64, 50000, 199874]),          # ragged, empty and tiny blocks
    (2.4e6, 48000.0, 700000, [300000, 400000]),                                 # 7-stage cascade
    (1.024e6, 96000.0, 300000, [100000, 200000]),                               # 7.68 = 1.92 x two half-band stages
    (1.024e6, 12500.0, 300000, [150000, 150000]),                               # rate 1: arbitrary resampler only
    (250000.0, 48000.0, 200000, [70000, 130000]),                               # shallow cascade, single-level front end
])
def test_pcm_within_one_lsb_of_oracle(fs, audio_rate, n, splits):
    from sdr_pmr446_amd import chain
    x = synth.synth_iq(n, fs, 1, dev_hz=2500.0)
    mb = max(splits)
    o = run(oracle.OracleDsd(fs_in=fs, audio_rate=audio_rate, max_block=mb), x, splits, ("pcm", "audio", "fm", "resampled"))
    g = run(chain.PmrDsd(fs_in=fs, audio_rate=audio_rate, max_block=mb), x, splits, ("pcm", "audio", "fm", "resampled"))
    assert g["n_out"] == o["n_out"]
    assert rel_err(g["resampled"], o["resampled"]) < 2e-5
    assert np.abs(g["fm"] - o["fm"]).max() < 3e-5
    assert np.abs(g["audio"] - o["audio"]).max() < 3e-5
    assert pcm_diff(g["pcm"], o["pcm"]).max() <= 1                              # north_star: int16 PCM within +-1 LSB
    assert np.abs(o["pcm"].astype(np.int32)).max() > 8000                       # ... of a signal that is really there


def test_split_invariance_and_reset():
    from sdr_pmr446_amd import chain
    n = 500000
    x = synth.synth_iq(n, 1.024e6, 1, dev_hz=1500.0)
    d = chain.PmrDsd(max_block=500000)
    a = run(d, x, [500000])
    d.reset()
    b = run(d, x, [123457, 200000, 1, 176542])
    assert len(a["pcm"]) == len(b["pcm"])
    assert pcm_diff(a["pcm"], b["pcm"]).max() <= 1
    assert np.abs(a["audio"] - b["audio"]).max() < 2e-5


def test_hip_matches_golden():
    from sdr_pmr446_amd import chain
    g, x = load_golden()
    d = chain.PmrDsd(fs_in=float(g["fs"]), max_block=int(max(g["splits"])))
    pcm = run(d, x, [int(s) for s in g["splits"]], ("pcm",))["pcm"]
    assert pcm.shape == g["pcm"].shape
    assert pcm_diff(pcm, g["pcm"]).max() <= 1


def test_device_resident_variant_and_capacity_errors():
    import torch
    from sdr_pmr446_amd import chain
    n = 200000
    x = synth.synth_iq(n, 1.024e6, 1, dev_hz=2500.0)
    ref = oracle.OracleDsd().process_block(x)
    d = chain.PmrDsd()
    dx = torch.from_numpy(x.view(np.float32).copy()).cuda()
    dp = torch.zeros(d.max_out, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()        # torch's fill runs on its own stream; the handle's stream is non-blocking
    nz = d.process_block_device(dx.data_ptr(), n, dp.data_ptr(), None, d.max_out)
    d.synchronize()
    assert nz == ref["n_out"]
    assert pcm_diff(dp[:nz].cpu().numpy(), ref["pcm"]).max() <= 1
    with pytest.raises(chain.PmrError):
        d.process_block_device(dx.data_ptr(), n, dp.data_ptr(), None, 10)       # cap < samples produced
    with pytest.raises(chain.PmrError):
        d.process_block(np.zeros(200001, np.complex64))                        # n_in > max_block
