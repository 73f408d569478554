"""Every rank of `bench.py --gpus N` runs ITS OWN synthetic stream (stream_id = rank: another noise realisation) -- cfg4 is
eight of them (BASELINE.json configs[3]; the reference's loop per stream, src/sdr_pmr446.c:788-908).  This file runs ALL EIGHT stream ids
on each of cfg2 / cfg3 / cfg5 (cfg3 x 8 = every GPU of cfg4) through the un-synchronised device entry against the oracle.

Bar: sdr_pmr446_amd/parity_rule.py, the rule bench.py's parity_check applies on every rank -- +-1 LSB on every PCM sample of every signal
channel, except where the response of the audio filter to an ILL-CONDITIONED discriminator sample reaches (discriminator inputs below 1 %
of the channel's steady rms in the oracle's channelizer output: the first frames after the reset, while the polyphase windows fill): there
the PCM must equal the oracle's plus what the MEASURED discriminator difference at those samples explains through the filter, within 2 LSB.
Round 5 used a blanket window (the first 409 frames of every channel, <= 8 LSB); round 6's first run of all 24 cases found stream 5 at cfg3
far outside it (37979 LSB: arg() of a channel output that is numerically zero) -- the 8-GPU run would have failed on rank 5.

On top of the PCM rule, the well-conditioned quantities are checked where the PCM rule is relaxed: the chain's channelizer outputs of
the start-up frames are within 1e-5 of the channel's scale of the oracle's, and the chain's discriminator output equals
arg(conj(r') r) / (2 pi kf) of ITS OWN channelizer outputs -- the difference to the oracle is the conditioning of arg(), not an error
in either step."""
import os

import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, active_channels
from sdr_pmr446_amd import parity_rule

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
    # the start-up through the debug taps: channelizer and discriminator outputs of the first frames (a second handle: the capture
    # selects the in-place carry form, bit-identical by tests/test_gpu_carry.py)
    nfirst = min(block, 1 << 22)
    gd = chain.PmrChain(fs_in=fs, num_channels=M, max_block=nfirst)
    rd = gd.process_block(x[:nfirst], want=("fm", "chan"))
    kf = gd.cfg.fm_kf
    gd.close()
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=1 << 22)
    outs = [o.process_block(x[p:p + (1 << 22)], want=("pcm", "chan", "fm")) for p in range(0, nblk * block, 1 << 22)]
    o.close()
    ref = np.concatenate([r["pcm"] for r in outs], axis=1).astype(np.int32)
    chan = np.concatenate([r["chan"] for r in outs], axis=1)
    act = active_channels(M, None, fs)
    F = rd["n_frames"]
    assert got.shape == ref.shape and got.shape[1] > 600 and F > parity_rule.PFB_FRAMES
    h = parity_rule.audio_response(*((parity_rule.fixtures(ROOT)[0], 4.0) + parity_rule.fixtures(ROOT)[1:]))
    v = parity_rule.check(got[act], ref[act], chan[act], rd["fm"][act], outs[0]["fm"][act, :F], h)
    ill = v.get("ill_conditioned")
    assert v["ok"], v
    # the ill-conditioned class is the start-up and nothing else
    assert 0 <= ill["last_frame"] < parity_rule.PFB_FRAMES, ill
    # ... and where it is, the well-conditioned quantities agree: channelizer outputs within 1e-5 of the channel's scale,
    cg, co = rd["chan"][act][:, :F].astype(np.complex128), chan[act][:, :F].astype(np.complex128)
    scale = np.sqrt((np.abs(chan[act][:, parity_rule.PFB_FRAMES:]) ** 2).mean(axis=1))[:, None]
    assert (np.abs(cg - co) / scale).max() <= 1e-5, float((np.abs(cg - co) / scale).max())
    # the chain's discriminator is arg(conj(r') r) / (2 pi kf) of ITS OWN channelizer outputs (first frame: r' = 0 -> 0)
    own = np.angle(np.conj(cg[:, :-1]) * cg[:, 1:]) / (2.0 * np.pi * kf)
    dd = np.abs(rd["fm"][act][:, 1:F].astype(np.float64) - own)
    dd = np.minimum(dd, np.abs(dd - 1.0 / kf))                                  # (+pi and -pi are the same angle)
    well = ~parity_rule.ill_conditioned(chan[act])[:, 1:F]
    assert dd[well].max() <= 2e-6, float(dd[well].max())
    assert np.median(dd[~well]) <= 1e-3 if (~well).any() else True      # f32 products of tiny numbers: loose, but the same angle
    assert np.abs(ref[act]).max() > 1000
