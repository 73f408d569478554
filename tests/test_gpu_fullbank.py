"""The FULL filter bank on the GPU: every one of the M channels synthesised, big un-synchronised blocks, int16 PCM
of the HIP chain vs the CPU oracle on EVERY channel that carries a signal -- the 256-channel plan all eight GPUs run in
cfg4 and the 1024-channel headline plan -- plus a floor under the agreement on the channels the +-1 LSB bar does not cover
(noise-only channels and the ones inside the chain's own dc-block notch, where arg() of a near-zero phasor is ill-conditioned).

Reference: the block loop src/sdr_pmr446.c:788-908 (one stream, state carried), channelizer :814, discriminator :881, audio
:882-904.  Buffers come from the library's own runtime (include/pmr_mem.h); the input is pmr_synth_iq_device's stream (all
channels, SURVEY s8d plan), downloaded once for the oracle.

And the dc-offset sweep that puts a measured boundary under the 2-LSB exception of tests/test_gpu_parity.py: from -40 dBFS of
DC (a healthy receiver) to -12 dBFS, the level where liquid's own float32 blocker state starts to carry more rounding noise
than one PCM LSB."""
import numpy as np
import pytest

from parity_util import CFG2, CFG3, CFG5, active_channels

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,log2_block,floor_rest", [(CFG2, 22, 0.85), (CFG3, 24, 0.85), (CFG5, 24, 0.85)], ids=["cfg2", "cfg3", "cfg5"])
def test_every_channel_loaded_pcm_within_one_lsb(cfg, log2_block, floor_rest):
    import oracle
    from sdr_pmr446_amd import chain, synth
    fs, M = cfg
    block, nblk = 1 << log2_block, 3
    iq = chain.synth_iq_device(nblk * block, fs, M, dev_hz=1500.0)               # ALL M channels, one stream of 3 blocks
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
    S = g.max_frames
    out = chain.DeviceBuffer(nblk * M * S * 2)
    chain.device_synchronize()
    ns = [g.process_block_device(iq.ptr + b * block * 8, block, d_pcm=out.ptr + b * M * S * 2, stride=S) for b in range(nblk)]
    g.synchronize()                                                              # (nothing synchronised in between)
    got = np.concatenate([out.download(np.int16, M * S, b * M * S * 2).reshape(M, S)[:, :ns[b]] for b in range(nblk)],
                         axis=1).astype(np.int32)
    x = iq.download(np.complex64, nblk * block)
    g.close(); iq.free(); out.free()
    chunk = 1 << 22
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=chunk)
    ref = np.concatenate([o.process_block(x[p:p + chunk], want=("pcm",))["pcm"] for p in range(0, len(x), chunk)],
                         axis=1).astype(np.int32)
    o.close()
    assert got.shape == ref.shape and got.shape[1] >= 3 * (block * M * 12500 // int(fs)) // M - 3
    act = active_channels(M, None, fs)
    rest = [k for k in range(M) if k not in set(act)]
    assert len(act) >= 0.85 * M and len(act) + len(rest) == M
    d = np.abs(got[act] - ref[act])
    assert d.max() <= 1, "channel %d differs by %d LSB" % (act[int(np.argmax(d.max(axis=1)))], d.max())
    assert np.abs(ref[act]).max() > 1000 and (np.abs(ref[act]).max(axis=1) > 100).all()      # every compared channel is alive
    # the channels outside the +-1 LSB bar: noise-only ones and those inside the dc-block notch.  No bit-level claim there, but a
    # regression confined to them (say, a wrong dc carry around band centre) must not pass unseen
    frac_rest = float((np.abs(got[rest] - ref[rest]) <= 1).mean())
    frac_all = float((np.abs(got - ref) <= 1).mean())
    print("within 1 LSB: %.4f of the %d excluded channels, %.4f overall" % (frac_rest, len(rest), frac_all))
    assert frac_rest >= floor_rest and frac_all >= 0.97


# (dc level re + j im, what must hold): -40 / -30 / -20 dBFS keep the +-1 LSB bar; -12 dBFS is the documented exception
DC_SWEEP = [(0.01 + 0.004j, 1, 1e-5), (0.03 + 0.012j, 1, 1e-5), (0.1 + 0.03j, 1, 3e-5), (0.25 + 0.1j, 2, 1e-4)]


@pytest.mark.parametrize("dc,pcm_tol,tol", DC_SWEEP, ids=["-40dBFS", "-30dBFS", "-20dBFS", "-12dBFS"])
def test_dc_offset_sweep_where_one_lsb_ends(dc, pcm_tol, tol):
    """A dc offset d parks liquid's direct-form-II blocker state at v ~ d / alpha = 2000 d: its float32 rounding noise per
    sample is ~ulp(v) / 2 -- 1e-6 at -40 dBFS, 3e-5 at -12 dBFS -- a realisation the scan-based GPU blocker does not share.
    Measured (MI355X, round 3): PCM stays within 1 LSB at all four levels on this signal; the float intermediates leave the 1e-5
    bar between -20 dBFS (6.7e-6) and -12 dBFS (1.6e-5), which is where the 2-LSB allowance of the -12 dBFS case comes from."""
    import oracle
    from sdr_pmr446_amd import chain, synth
    fs, M = CFG2
    n = 400000
    x = synth.synth_iq(n, fs, M, dev_hz=500.0, dc_offset=dc)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n // 2)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n // 2)
    worst, worst_rs = 0, 0.0
    for b in range(2):
        ro = o.process_block(x[b * n // 2:(b + 1) * n // 2], want=("pcm", "resampled"))
        rg = g.process_block(x[b * n // 2:(b + 1) * n // 2], want=("pcm", "resampled"))
        act = active_channels(M)
        worst = max(worst, int(np.abs(rg["pcm"][act].astype(np.int32) - ro["pcm"][act].astype(np.int32)).max()))
        worst_rs = max(worst_rs, float(np.abs(rg["resampled"] - ro["resampled"]).max() / np.abs(ro["resampled"]).max()))
    o.close(); g.close()
    print("dc %.3f: PCM max diff %d LSB, resampled rel err %.2e" % (abs(dc), worst, worst_rs))
    assert worst <= pcm_tol and worst_rs < tol
