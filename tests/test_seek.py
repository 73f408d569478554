"""A stream never ends (the reference's `while(!exit_via_sig)`, src/sdr_pmr446.c:788): its counters pass 2^32 after 4.3 s at cfg5
(n_raw), 70 s at cfg3.  No test can stream there through the CPU oracle, so both sides get a SEEK -- "the state after n_raw zero
samples": every filter state zero, the closed-form counters (msresamp buffer index, resampler phase, ring remainder, NCO phase,
the detector's block count) where that many zeros would have left them.

CPU tier (this file's un-marked tests): the oracle's seek equals really feeding the zeros, bit for bit, and the library's
host-side plan (pmr_cfg_plan_block) arrives at the same counts.  GPU tier: the library's seek equals feeding zeros, bit for bit;
then ragged un-synchronised blocks ACROSS n_raw = 2^32, xr_abs = 2^32, frames x M = 2^32 and a block whose decimated index passes
2^24, +-1 LSB against the oracle seeked to the same place."""
import ctypes as C

import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, active_channels
from sdr_pmr446_amd import synth


def _run(c, x, splits, want=("pcm",)):
    outs, pos = {}, 0
    for n in splits:
        o = c.process_block(x[pos:pos + n], want=want)
        pos += n
        for k in want:
            outs.setdefault(k, []).append(o[k])
    return {k: np.concatenate(v, axis=-1) for k, v in outs.items()}      # (resampled: 1-D; everything else [M][time])


@pytest.mark.parametrize("cfg,n0", [(CFG2, 1234567), (CFG2, 8 * 2441 * 16 * 12 + 5), (CFG3, 777777), ((1.024e6, 16), 300001)],
                         ids=["cfg2", "cfg2-past-ctcss-blocks", "cfg3", "reference-plan"])
def test_oracle_seek_equals_feeding_zeros(cfg, n0):
    fs, M = cfg
    n = 150000 if M <= 16 else 1 << 19
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0, channels=list(range(0, M, max(1, M // 16))))
    splits = [n // 3, n - n // 3 - 7, 7]
    mb = max(max(splits), 1 << 18)
    a = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb)
    z = np.zeros(mb, dtype=np.complex64)
    left = n0
    while left:
        k = min(left, mb)
        a.process_block(z[:k], want=("pcm",))
        left -= k
    ra = _run(a, x, splits, ("pcm", "ctcss", "resampled")) if M <= 16 else _run(a, x, splits, ("pcm",))
    b = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb)
    b.seek(n0)
    rb = _run(b, x, splits, ("pcm", "ctcss", "resampled")) if M <= 16 else _run(b, x, splits, ("pcm",))
    a.close(); b.close()
    for k in ra:
        assert ra[k].shape == rb[k].shape and ra[k].tobytes() == rb[k].tobytes(), k
    assert ra["pcm"].shape[1] > 100 and np.abs(ra["pcm"]).max() > 1000


def test_host_plan_agrees_with_the_oracle_seek_at_large_positions():
    """pmr_cfg_plan_block (the closed form the library sizes its launches with, no device needed) fed ONE block ending at n0 -- in
    pieces, the counters are 64-bit -- against the frame counts of the oracle seeked there."""
    from sdr_pmr446_amd import chain
    L = chain.load()
    for (fs, M), n0 in [(CFG2, (1 << 32) - 12345), (CFG2, 12 * (1 << 32) + 99), (CFG3, (1 << 34) + 4321), (CFG5, (1 << 36) + 17)]:
        cfg = chain.make_cfg(fs_in=fs, num_channels=M, max_block=1 << 20)
        st = chain.PlanState()
        left, tot_ny = n0, 0
        ny, ns = C.c_uint(0), C.c_uint(0)
        while left:
            k = min(left, 1 << 31)
            assert L.pmr_cfg_plan_block(C.byref(cfg), C.byref(st), k, C.byref(ny), C.byref(ns)) == 0
            tot_ny += ny.value
            left -= k
        o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=1 << 20)
        o.seek(n0)
        n = 300001
        x = synth.synth_iq(n, fs, M, dev_hz=1500.0, channels=[1, M // 2 + 1])
        r = o.process_block(x, want=("pcm", "resampled"))
        o.close()
        assert L.pmr_cfg_plan_block(C.byref(cfg), C.byref(st), n, C.byref(ny), C.byref(ns)) == 0
        assert (ny.value, ns.value) == (len(r["resampled"]), r["n_frames"]), (fs, M, n0)
        assert st.n_raw == n0 + n and tot_ny > n0 * (M * 12500.0 / fs) * 0.999


# ------------------------------------------------------------------------------------------------------------------ GPU tier

def _synth_dev(n, fs, M, **kw):
    from sdr_pmr446_amd import chain
    buf = chain.synth_iq_device(n, fs, M, **kw)
    x = buf.download(np.complex64, n)
    buf.free()
    return x


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n0,ctcss", [(CFG2, 1234567, True), (CFG3, 7777777, False), (CFG5, (1 << 24) + 12345, False)],
                         ids=["cfg2-ctcss", "cfg3", "cfg5"])
def test_gpu_seek_equals_feeding_zeros(cfg, n0, ctcss):
    """pmr_chain_seek against the same handle type really fed n0 zero samples: PCM, audio and CTCSS events bit for bit."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    n = 400000 if M <= 16 else 1 << 22
    x = _synth_dev(n, fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0)
    splits = [n // 2 + 11, n - n // 2 - 11 - 3, 3]
    mb = max(splits[0], 1 << 20)
    want = ("pcm", "audio") + (("ctcss",) if ctcss else ())
    res = []
    for seek in (False, True):
        g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
        if ctcss:
            g.ctcss_enable()
        if seek:
            g.seek(n0)
        else:
            z = np.zeros(mb, dtype=np.complex64)
            left = n0
            while left:
                k = min(left, mb)
                g.process_block(z[:k], want=("pcm",))
                left -= k
        pos0 = g.position()
        res.append((_run(g, x, splits, want), pos0, g.position()))
        g.close()
    (a, pa0, pa1), (b, pb0, pb1) = res
    assert pa0 == pb0 and pa1 == pb1 and pa0[0] == n0 and pa1[0] == n0 + n
    for k in want:
        assert a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), k
    assert np.abs(a["pcm"]).max() > 1000


def _across(cfg, n0, n, splits, ctcss=False, atol_lsb=1, mb=None, pipelined=True):
    """Handle and oracle seeked to n0, the same ragged blocks through both; the GPU side by un-synchronised device calls."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    assert sum(splits) == n
    mb = mb or max(splits)
    iq = chain.synth_iq_device(n, fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0)
    x = iq.download(np.complex64, n)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
    if ctcss:
        g.ctcss_enable()
    g.seek(n0)
    pos_start = g.position()
    S = g.max_frames
    bufs = [chain.DeviceBuffer(M * S * 2) for _ in splits]
    ns, pos, evs = [], 0, []
    for i, k in enumerate(splits):
        ns.append(g.process_block_device(iq.ptr + pos * 8, k, d_pcm=bufs[i].ptr, stride=S))
        pos += k
        if ctcss:
            evs.append(g.ctcss_read())                         # (synchronises: the events of THIS block)
    g.synchronize()
    got = np.concatenate([bufs[i].download(np.int16, M * S).reshape(M, S)[:, :ns[i]] for i in range(len(splits))], axis=1)
    pos_end = g.position()
    g.close(); iq.free()
    for b in bufs:
        b.free()
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb)
    o.seek(n0)
    ro = _run(o, x, splits, ("pcm", "ctcss") if ctcss else ("pcm",))
    o.close()
    act = active_channels(M, None, fs)
    assert got.shape == ro["pcm"].shape and got.shape[1] > 50
    d = np.abs(got[act].astype(np.int32) - ro["pcm"][act].astype(np.int32))
    assert d.max() <= atol_lsb, (int(d.max()), float((d > 1).mean()))
    assert np.abs(ro["pcm"][act]).max() > 1000
    if ctcss:
        ge = np.concatenate(evs, axis=1)
        assert ge.shape == ro["ctcss"].shape and ge.shape[1] >= 2
        for k in [c for c in act if synth.channel_kind(c) == "fm"]:
            assert np.array_equal(ge["index"][k], ro["ctcss"]["index"][k])
            assert np.array_equal(ge["detected"][k], ro["ctcss"]["detected"][k])
    return pos_start, pos_end


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["n_raw", "xr_abs", "frames"])
@pytest.mark.parametrize("cfg", [CFG2, CFG3, CFG5], ids=["cfg2", "cfg3", "cfg5"])
def test_blocks_across_two_to_the_32(cfg, what):
    """Ragged blocks whose raw-sample count / resampled-sample count / FRAME count passes 2^32 in the middle of a block (cfg5 reaches
    the first after 4.3 s of streaming, the last after four days; cfg3 / cfg5: un-synchronised device calls; cfg2: CTCSS detector on,
    whose 2441-frame grid comes from the frame count -- reading its events synchronises)."""
    fs, M = cfg
    rate = M * 12500.0 / fs
    n = (1 << 21) if M <= 16 else (1 << 23)
    splits = [n // 4 + 5, n // 2 - 77, n - (n // 4 + 5) - (n // 2 - 77)]
    idx = {"n_raw": 0, "xr_abs": 1, "frames": 2}[what]
    target = int((1 << 32) / (1.0, rate, rate / M)[idx])
    n0 = target - n // 2 - 1234                                # the crossing falls inside the second block
    pos0, pos1 = _across(cfg, n0, n, splits, ctcss=(M <= 16))
    assert pos0[idx] < (1 << 32) < pos1[idx], (pos0, pos1)


@pytest.mark.gpu
def test_one_block_whose_decimated_index_passes_two_to_the_24():
    """cfg2, ONE call of 2^27 + 12345 samples (1 GB of cf32): the block's decimated-sample index (the resampler's `phase >> 24`,
    2^24 per decimated sample) and its tile-local bookkeeping pass 2^24 inside the call; stream origin beyond 2^32."""
    n = (1 << 27) + 12345
    _across(CFG2, (1 << 33) + 7, n, [n], ctcss=False, mb=n)


@pytest.mark.gpu
@pytest.mark.nopoison
def test_streaming_past_two_to_the_32_at_cfg5_equals_a_handle_seeked_there():
    """The soak entry: 66 un-synchronised 2^26-sample blocks at cfg5 (4.4e9 samples, n_raw passes 2^32 in block 64) through one
    handle; the blocks after that against a second handle seeked to the same position.  The seeked handle starts from zero state,
    so its first block still carries start-up transients (dc blocker: 1 / alpha = 2000 samples; audio FIR: 383 frames); from the
    second block on the two agree within 1 LSB."""
    from sdr_pmr446_amd import chain
    fs, M = CFG5
    lb, rot = 26, 4
    block = 1 << lb
    iq = chain.synth_iq_device(rot * block, fs, M, period_log2=lb + 2)
    a = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
    S = a.max_frames
    bufs = [chain.DeviceBuffer(M * S * 2) for _ in range(3)]
    nb = 66
    for b in range(nb):
        a.process_block_device(iq.ptr + (b % rot) * block * 8, block, d_pcm=bufs[0].ptr, stride=S)
    assert a.position()[0] == nb * block > (1 << 32)
    s = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block)
    s.seek(nb * block)
    assert s.position() == a.position()
    out = []
    for c in (a, s):
        ns = [c.process_block_device(iq.ptr + ((nb + i) % rot) * block * 8, block, d_pcm=bufs[1 + i].ptr, stride=S) for i in range(2)]
        c.synchronize()
        out.append(bufs[2].download(np.int16, M * S).reshape(M, S)[:, :ns[1]].astype(np.int32))
    a.close(); s.close(); iq.free()
    for b in bufs:
        b.free()
    act = active_channels(M, None, fs)
    d = np.abs(out[0][act] - out[1][act])
    assert out[0].shape[1] > 700 and np.abs(out[0][act]).max() > 1000
    assert d.max() <= 1, (int(d.max()), float((d > 1).mean()))


@pytest.mark.gpu
def test_seek_range_limit_and_position_semantics():
    """include/pmr_chain.h: n_raw >= 2^62 is PMR_ERANGE and leaves the handle where it was; pmr_chain_position counts ENQUEUED blocks
    (neither was stated nor tested)."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n = 1 << 18
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
    g.seek(12345)
    assert g.position()[0] == 12345
    assert g._L.pmr_chain_seek(g.h, 1 << 62) == 2                      # PMR_ERANGE
    assert b"seek position" in g._L.pmr_chain_last_error(g.h)
    assert g.position()[0] == 12345                                   # untouched
    assert g._L.pmr_chain_seek(g.h, (1 << 62) - 1) == 0 and g.position()[0] == (1 << 62) - 1
    g.seek(0)
    iq = chain.synth_iq_device(n, fs, M)
    S = g.max_frames
    pcm = chain.DeviceBuffer(M * S * 2)
    ns = g.process_block_device(iq.ptr, n, d_pcm=pcm.ptr, stride=S)   # not synchronised: the position is the planned one, already
    raw, res, frames = g.position()
    assert raw == n and frames == ns and res >= frames * M
    g.synchronize()
    assert g.position() == (raw, res, frames)
    g.close(); iq.free(); pcm.free()
