"""Every rank of `bench.py --gpus N` runs ITS OWN synthetic stream (stream_id = rank: another noise realisation) -- cfg4 is
eight of them (BASELINE.json configs[3]; the reference's loop per stream, src/sdr_pmr446.c:788-908).  This file runs ALL EIGHT stream ids
on each of cfg2 / cfg3 / cfg5 (cfg3 x 8 = every GPU of cfg4) through the un-synchronised device entry against the oracle.

Bar: sdr_pmr446_amd/parity_rule.py, the rule bench.py's parity_check applies on every rank -- +-1 LSB on every sample of every signal
channel, except PCM samples the audio FIR connects to an ILL-CONDITIONED discriminator sample (inputs below 1 % of the channel's steady
rms in the oracle's channelizer output: the first frames after the reset, while the polyphase windows fill): <= 4 LSB at the FIR's centre
lags, <= 2 LSB at its other lags.  Round 5 used a blanket window (the first 409 frames of every channel, <= 8 LSB); its one known case
(stream 1 at cfg5: 3 LSB at channel 410, frame 200 = ill frame 10 + lag 190, profiles/r05_stream_parity.txt) sits in the centre class.
The number of samples that USE the relaxation (> 1 LSB) is asserted to stay in single digits per stream."""
import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, active_channels
from sdr_pmr446_amd import parity_rule

pytestmark = pytest.mark.gpu

CASES = [(name, cfg, sid, lb, nblk) for name, cfg, lb, nblk in (("cfg5", CFG5, 25, 2), ("cfg3", CFG3, 23, 2), ("cfg2", CFG2, 20, 3))
         for sid in range(8)]


@pytest.mark.parametrize("name,cfg,sid,lb,nblk", CASES, ids=["%s-stream%d" % (c[0], c[2]) for c in CASES])
def test_every_ranks_stream_matches_the_oracle(name, cfg, sid, lb, nblk):
    from sdr_pmr446_amd import chain
    fs, M = cfg
    block = 1 << lb
    iq = chain.synth_iq_device(nblk * block, fs, M, stream_id=sid, period_log2=28)       # bench.py's stream: rotation period 2^28
    x = iq.download(np.complex64, nblk * block)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
    S = g.max_frames
    bufs = [chain.DeviceBuffer(M * S * 2) for _ in range(nblk)]
    ns = [g.process_block_device(iq.ptr + b * block * 8, block, d_pcm=bufs[b].ptr, stride=S) for b in range(nblk)]
    g.synchronize()
    got = np.concatenate([bufs[b].download(np.int16, M * S).reshape(M, S)[:, :ns[b]] for b in range(nblk)], axis=1).astype(np.int32)
    g.close(); iq.free()
    for b in bufs:
        b.free()
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=1 << 22)
    outs = [o.process_block(x[p:p + (1 << 22)], want=("pcm", "chan")) for p in range(0, nblk * block, 1 << 22)]
    o.close()
    ref = np.concatenate([r["pcm"] for r in outs], axis=1).astype(np.int32)
    chan = np.concatenate([r["chan"] for r in outs], axis=1)
    act = active_channels(M, None, fs)
    assert got.shape == ref.shape and got.shape[1] > parity_rule.PFB_FRAMES + parity_rule.FIR_TAPS + 200
    v = parity_rule.check(got[act], ref[act], chan[act])
    ill = v["ill_conditioned"]
    assert v["ok"], v
    # the ill-conditioned class is the start-up and nothing else, and hardly any sample needs the relaxation
    assert 0 <= ill["last_frame"] < parity_rule.PFB_FRAMES, ill
    assert ill["samples_over_1_lsb"] <= 8, ill
    assert np.abs(ref[act]).max() > 1000
