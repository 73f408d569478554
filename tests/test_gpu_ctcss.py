"""SURVEY s8 row f2 on the GPU: CTCSS tone detection for all channels vs the oracle restatement of
src/sdr_pmr446.c:338-418, :605-628 -- tone index and decision exact where the decision is clear, powers within 0.5 %
(the reference's float32 Goertzel recurrence itself wobbles by ~1e-5), Goertzel blocks that straddle calls included."""
import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, active_channels
from sdr_pmr446_amd import synth

pytestmark = pytest.mark.gpu


def _run(chain_obj, x, splits):
    evs, pos = [], 0
    for n in splits:
        o = chain_obj.process_block(x[pos:pos + n], want=("pcm", "ctcss"))
        pos += n
        evs.append(o["ctcss"])
    return np.concatenate(evs, axis=1)


@pytest.mark.parametrize("fs,M,N,splits,ks", [
    CFG2 + (1300000, [1300000], None),
    CFG2 + (1300000, [400000, 1, 500000, 399999], None),          # 2441-frame Goertzel blocks straddle the calls
    CFG3 + (1 << 25, [1 << 24, 1 << 24], list(range(0, 256, 9))),
    CFG2 + (1 << 25, [(1 << 24) + 12345, (1 << 24) - 12345], None),   # > 64 Goertzel blocks per call: several blocks per workgroup
], ids=["cfg2-one-call", "cfg2-ragged", "cfg3-2^24", "cfg2-2^24-blocks"])
def test_ctcss_matches_oracle(fs, M, N, splits, ks):
    from sdr_pmr446_amd import chain
    x = synth.synth_iq(N, fs, M, channels=ks, dev_hz=1500.0, ctcss_dev_hz=700.0)
    mb = max(splits)
    eo = _run(oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb), x, splits)
    eg = _run(chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb), x, splits)
    assert eo.shape == eg.shape and eo.shape[1] >= 2
    fm_ch = [k for k in active_channels(M, ks) if synth.channel_kind(k) == "fm"]
    for k in fm_ch:
        assert np.all(eo["detected"][k] == 1)                                   # the test signal is a clear tone
        assert np.array_equal(eg["index"][k], eo["index"][k])
        assert np.all(eo["index"][k] == k % 38)                                 # ... and it is the right tone
        assert np.array_equal(eg["detected"][k], eo["detected"][k])
        assert np.allclose(eg["max_power"][k], eo["max_power"][k], rtol=5e-3)
        assert np.allclose(eg["avg_power"][k], eo["avg_power"][k], rtol=5e-3)
    carriers = [k for k in active_channels(M, ks) if synth.channel_kind(k) == "carrier"]
    for k in carriers:                                                          # bare carrier: no tone, in both
        assert not eo["detected"][k].any() and not eg["detected"][k].any()


@pytest.mark.parametrize("opts,masked", [({}, False), ({}, True), (dict(lowpass=True), False)],
                         ids=["dual-pass", "dual-pass-masked", "two-passes-lowpass"])
def test_ctcss_low_pass_branch_matches_oracle(opts, masked):
    """The intermediate the detector runs on -- tmp1 = delay188(x) - hp(x), reference src/sdr_pmr446.c:884-889 -- against the
    oracle's, sample by sample.  With the default audio chain it comes out of the SAME MFMA pass as the audio (second tap set);
    with follow-on FIR passes out of a pass of its own."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n, splits = 260000, [100000, 60001, 99999]
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(splits), **opts)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(splits), **opts)
    en = [0, 2, 5, 9] if masked else list(range(M))
    if masked:
        g.set_channel_mask(en)
    pos, frames, checked = 0, 0, 0
    for b, s in enumerate(splits):
        ro = o.process_block(x[pos:pos + s], want=("pcm", "ctcss_lp", "ctcss"))
        rg = g.process_block(x[pos:pos + s], want=("pcm", "ctcss_lp", "ctcss"))
        pos += s
        act = [k for k in active_channels(M) if k in en]
        skip = max(0, 700 - frames)                       # start-up: |chan| ~ 0 makes arg() ill-conditioned and rings through the FIR
        frames += ro["n_frames"]
        assert rg["ctcss_lp"].shape == ro["ctcss_lp"].shape
        d = np.abs(rg["ctcss_lp"][act][:, skip:] - ro["ctcss_lp"][act][:, skip:])
        checked += d.size
        assert d.size == 0 or d.max() < 2e-6, d.max()
        assert np.abs(rg["pcm"][act].astype(np.int32) - ro["pcm"][act].astype(np.int32)).max() <= 1
        assert np.array_equal(rg["ctcss"]["index"][act], ro["ctcss"]["index"][act])
    assert checked > 2000


def test_ctcss_decisions_around_the_thresholds():
    """Tone levels swept THROUGH the decision thresholds avg > 120 && max/avg > 10 (reference src/sdr_pmr446.c:403-404): the
    powers agree with the oracle within 0.5 %, and the decision agrees wherever the oracle's own margin to a threshold
    exceeds that tolerance (inside the margin either answer is a rounding coin-flip in the reference itself)."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    n, splits = 2400000, [900000, 600001, 899999]
    # Hz of CTCSS deviation: avg power crosses 120 near 300 Hz
    devs = {k: d for k, d in zip(range(M), [150, 200, 230, 245, 250, 252, 255, 260, 280, 300, 320, 340, 400, 500, 600, 700])}
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0, ctcss_dev_of=lambda k: devs[k])
    eo = _run(oracle.OracleChain(fs_in=fs, num_channels=M, max_block=max(splits)), x, splits)
    eg = _run(chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(splits)), x, splits)
    assert eo.shape == eg.shape and eo.shape[1] >= 4
    fm_ch = [k for k in active_channels(M) if synth.channel_kind(k) == "fm"]
    tol = 5e-3
    seen = {0: 0, 1: 0}
    near = 0
    for k in fm_ch:
        assert np.allclose(eg["avg_power"][k], eo["avg_power"][k], rtol=tol)
        assert np.allclose(eg["max_power"][k], eo["max_power"][k], rtol=tol)
        for b in range(1, eo.shape[1]):                       # (block 0 holds the start-up transient)
            avg, mx = float(eo["avg_power"][k, b]), float(eo["max_power"][k, b])
            margin = min(abs(avg - 120.0) / 120.0, abs(mx / avg - 10.0) / 10.0) if avg > 0 else 1.0
            if margin > 2 * tol:
                assert eg["detected"][k, b] == eo["detected"][k, b], (k, b, avg, mx / avg)
                seen[int(eo["detected"][k, b])] += 1
            else:
                near += 1
    assert seen[0] >= 4 and seen[1] >= 4, (seen, near)        # the sweep really straddles the avg > 120 threshold (252 Hz ~ 120)


def test_ctcss_decisions_equal_the_references_own_detector_code():
    """The HIP detector against decisions taken by the REFERENCE's ctcss_detector_analyze (src/sdr_pmr446.c:366-409, compiled from
    the reference in the build container: tests/golden/ctcss_ref.npz, tools/make_ref_fixtures.py) on the same synthetic tone-level
    sweep: tone index and decision identical wherever the reference's own margin to a threshold (:403-404) exceeds the 0.5 %
    power tolerance, strongest-tone power within that tolerance."""
    import os
    from sdr_pmr446_amd import chain
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ctcss_ref.npz"))
    fs, M, n = float(g["synth_fs"]), int(g["synth_M"]), int(g["synth_n"])
    devs = list(g["synth_ctcss_devs"])
    x = synth.synth_iq(n, fs, M, dev_hz=float(g["synth_dev_hz"]), ctcss_dev_of=lambda k: devs[k])
    splits = [900000, 600001, 899999]
    eg = _run(chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(splits)), x, splits)
    B = g["index"].shape[1]
    assert eg.shape == (M, B)
    tol, seen = 5e-3, {0: 0, 1: 0}
    for k in [k for k in active_channels(M) if synth.channel_kind(k) == "fm"]:
        for b in range(1, B):                                 # (block 0 holds the start-up transient)
            mx = float(g["max_power"][k, b])
            avg = float(np.float32(g["power"][k, b].astype(np.float32).sum(dtype=np.float32) / np.float32(38)))
            assert eg["max_power"][k, b] == pytest.approx(mx, rel=tol)
            margin = min(abs(avg - 120.0) / 120.0, abs(mx / avg - 10.0) / 10.0)
            if margin > 2 * tol:
                assert eg["index"][k, b] == g["index"][k, b] and eg["detected"][k, b] == g["detected"][k, b], (k, b, avg, mx / avg)
                seen[int(g["detected"][k, b])] += 1
    assert seen[0] >= 4 and seen[1] >= 4, seen


@pytest.mark.parametrize("cfg,k,n,splits", [(CFG2, 5, 1300000, [400000, 1, 500000, 399999]),
                                             (CFG3, 100, 1 << 25, [1 << 24, 1 << 24]),
                                             (CFG2, 9, 1 << 25, [1 << 25])], ids=["cfg2-ch5", "cfg3-ch100", "cfg2-ch9-72-blocks"])
def test_ctcss_with_one_open_channel_is_the_references_mode(cfg, k, n, splits):
    """The reference runs ctcss_execute for the squelch-selected channel only (src/sdr_pmr446.c:893).  mask = {k}  <->
    OracleChain(only_channel=k): the open channel's events as in the all-channel comparison; every closed channel reports
    {index -1, detected 0} (the detector kernels never touched it)."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    ks = sorted(set([k] + list(range(0, M, max(1, M // 16)))))
    x = synth.synth_iq(n, fs, M, channels=ks, dev_hz=1500.0, ctcss_dev_hz=700.0)
    mb = max(splits)
    eo = _run(oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb, only_channel=k), x, splits)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
    g.set_channel_mask([k])
    eg = _run(g, x, splits)
    assert eo.shape == eg.shape and eo.shape[1] >= 2
    assert np.all(eo["detected"][k] == 1) and np.all(eo["index"][k] == k % 38)
    assert np.array_equal(eg["index"][k], eo["index"][k]) and np.array_equal(eg["detected"][k], eo["detected"][k])
    assert np.allclose(eg["max_power"][k], eo["max_power"][k], rtol=5e-3)
    assert np.allclose(eg["avg_power"][k], eo["avg_power"][k], rtol=5e-3)
    closed = [c for c in range(M) if c != k]
    assert np.all(eg["index"][closed] == -1) and not eg["detected"][closed].any()
    # open a second channel in mid-stream: its detector starts from zero sums at that point, the first channel is unaffected
    g.set_channel_mask([k, ks[1] if ks[1] != k else ks[2]])
    e2 = g.process_block(x[:splits[0]], want=("pcm", "ctcss"))["ctcss"]
    assert e2.shape[0] == M


def _feed_until(chains, x, pos, first, target_frames):
    """Feed every chain the same blocks: `first` samples, then single frames' worth (192 samples at cfg2) until `target_frames`
    frames have been channelized.  Returns (new position, [events per chain])."""
    evs = [[] for _ in chains]
    done = None
    n = first
    while done is None or done < target_frames:
        nf = []
        for i, c in enumerate(chains):
            o = c.process_block(x[pos:pos + n], want=("pcm", "ctcss"))
            evs[i].append(o["ctcss"]); nf.append(o["n_frames"])
        assert len(set(nf)) == 1
        done = nf[0] if done is None else done + nf[0]
        pos += n
        n = 192
    assert done == target_frames
    return pos, evs


@pytest.mark.parametrize("extra_frames", [0, 700], ids=["on-a-block-boundary", "mid-block"])
def test_reset_channel_with_the_detector_on(extra_frames):
    """pmr_chain_reset_channel with the CTCSS detector running, against the oracle's freqdem_reset + ctcss_detector_reset
    (reference src/sdr_pmr446.c:866-867).  The reference does NOT reset ctcss_dcblock (:606): every channel here
    carries a 300 Hz carrier offset, i.e. dc in the discriminator output, so a wrongly zeroed blocker state would put a step
    transient into the low Goertzel bins of the next block.
    * reset on a Goertzel-block boundary: the GPU's shared 2441-frame grid and the oracle's restarted count coincide -- every event
      of every channel, the reset one's first block included, equals the oracle's (powers within 0.5 %);
    * reset in mid-block: the block in progress is incomplete for that channel -> {index -1, detected 0} (documented deviation:
      the reference's single detector restarts its own block count instead, include/pmr_chain.h); the blocks after it decide like
      the oracle's; every other channel is unaffected."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    N, k = 2441, 5
    n = (5 * N + 1500) * 192
    t = np.arange(n, dtype=np.float64) / fs
    x = (synth.synth_iq(n, fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0) * np.exp(2j * np.pi * 300.0 * t)).astype(np.complex64)
    mb = 3 * N * 192
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
    o_all = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb)
    o_k = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=mb, only_channel=k)
    chains = [g, o_all, o_k]
    pos, ev1 = _feed_until(chains, x, 0, (2 * N - 30) * 192, 2 * N + extra_frames)
    for c in chains:
        c.reset_channel(k)
    rest = [(N + 400) * 192, len(x) - pos - (N + 400) * 192]
    ev2 = [[] for _ in chains]
    for nb in rest:
        for i, c in enumerate(chains):
            ev2[i].append(c.process_block(x[pos:pos + nb], want=("pcm", "ctcss"))["ctcss"])
        pos += nb
    eg1, eo1 = np.concatenate(ev1[0], axis=1), np.concatenate(ev1[1], axis=1)
    eg2, eo2, ek2 = (np.concatenate(e, axis=1) for e in ev2)
    for c in chains:
        c.close()
    fm_ch = [c for c in active_channels(M) if synth.channel_kind(c) == "fm"]
    assert k in fm_ch and eg1.shape == eo1.shape and eg1.shape[1] == 2
    others = [c for c in fm_ch if c != k]

    def same(a, b, chans):
        for c in chans:
            assert np.array_equal(a["index"][c], b["index"][c]) and np.array_equal(a["detected"][c], b["detected"][c]), c
            assert np.allclose(a["max_power"][c], b["max_power"][c], rtol=5e-3), c
            assert np.allclose(a["avg_power"][c], b["avg_power"][c], rtol=5e-3), c
    same(eg1, eo1, fm_ch)                                     # before the reset
    assert eg2.shape[1] >= 2
    same(eg2, eo2[:, :eg2.shape[1]], others)                  # the other channels never notice
    if extra_frames == 0:
        assert ek2.shape[1] == eg2.shape[1]
        same(eg2, ek2, [k])                                   # first block after the reset included: blocker state kept, sums from zero
        assert np.all(ek2["detected"][k] == 1) and np.all(ek2["index"][k] == k % 38)
    else:
        assert eg2["index"][k, 0] == -1 and eg2["detected"][k, 0] == 0 and eg2["max_power"][k, 0] == 0.0
        assert np.all(eg2["index"][k, 1:] == k % 38) and np.all(eg2["detected"][k, 1:] == 1)
        assert np.all(ek2["index"][k] == k % 38) and np.all(ek2["detected"][k] == 1)      # ... as the reference decides on its own grid


def test_reopened_channels_restart_and_events_follow_the_mask_of_their_block():
    """(1) set_channel_mask(NULL) re-opens every channel -- the frozen partial Goertzel sums of the channels that were
    closed must restart from zero like in the explicit-list path (their first, incomplete block reports no decision; afterwards
    they decide like the oracle); (2) pmr_chain_ctcss_read reports a block's events under the mask THAT BLOCK ran with, not the
    mask set since (reference: ctcss_execute runs for the active channel of the block in hand, src/sdr_pmr446.c:893)."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    N, k = 2441, 5
    n = (6 * N) * 192
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=3 * N * 192)
    fm_ch = [c for c in active_channels(M) if synth.channel_kind(c) == "fm"]
    g.set_channel_mask([k])
    n1 = (N + 900) * 192                                      # ends in mid-block
    r1 = g.process_block(x[:n1], want=("pcm",))
    g.ctcss_enable(True)
    # (the detector was off for the first block: switch it on and run a block under the one-channel mask)
    n2 = N * 192
    g.process_block(x[n1:n1 + n2], want=("pcm", "ctcss"))
    g.set_channel_mask([9])                                   # BEFORE reading: the events still describe the block run under {k}
    ev = g.ctcss_read()
    assert ev.shape[1] >= 1
    assert ev["index"][k, -1] == k % 38 and ev["detected"][k, -1] == 1
    assert np.all(ev["index"][[c for c in range(M) if c != k]] == -1)
    g.set_channel_mask([k])
    g.process_block(x[n1 + n2:n1 + 2 * n2], want=("pcm", "ctcss"))      # channel 9 closed again, k reopened in mid-block
    g.set_channel_mask(None)                                  # everything opens, in mid-block
    e3 = g.process_block(x[n1 + 2 * n2:n1 + 2 * n2 + 2 * N * 192 + 5000], want=("pcm", "ctcss"))["ctcss"]
    g.close()
    assert e3.shape[1] >= 2
    for c in fm_ch:
        if c == k:
            continue
        assert e3["index"][c, 0] == -1 and e3["detected"][c, 0] == 0, c        # incomplete block: no decision, and no stale sums
        assert np.all(e3["index"][c, 1:] == c % 38) and np.all(e3["detected"][c, 1:] == 1), c
    assert np.all(np.isfinite(e3["max_power"])) and np.all(np.isfinite(e3["avg_power"]))
    assert r1["n_frames"] > N


@pytest.mark.parametrize("first_frames,expect_first", [(2441, "decision"), (2441 // 2 + 1, "none")],
                         ids=["audio-part-starts-on-the-grid", "audio-part-starts-in-mid-block"])
def test_two_step_mask_change_restarts_relative_to_the_pending_audio_part(first_frames, expect_first):
    """in the two-step form (pmr_chain_channelize_block -> set_channel_mask -> pmr_chain_demodulate_block, the squelch
    order of the reference, src/sdr_pmr446.c:828-877) the frame counter has already advanced past the pending block when the mask
    opens a channel, but the detector runs NEXT at the pending block's first frame: "restarted in mid-block" must be judged there.
    Pending part starting ON the 2441-frame grid: the channel sees the whole Goertzel block -> a real decision (the old code
    reported {-1, 0, 0, 0}); starting in mid-block: that block is incomplete for the channel -> no decision (the old code reported
    a decision over a partial block whenever the advanced counter happened to sit on the grid)."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    N, k = 2441, 5
    fm_ch = [c for c in active_channels(M) if synth.channel_kind(c) == "fm" and c != k]
    c = fm_ch[0]
    second_frames = 2 * N - first_frames if expect_first == "none" else N + N // 2
    n1, n2 = first_frames * 192, second_frames * 192           # cfg2: exactly 192 raw samples per frame
    x = synth.synth_iq(n1 + n2, fs, M, dev_hz=1500.0, ctcss_dev_hz=700.0)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(n1, n2))
    g.ctcss_enable(True)
    g.set_channel_mask([k])
    assert g.process_block(x[:n1], want=("pcm",))["n_frames"] == first_frames
    assert g.channelize_block(x[n1:], want=("rssi",))["n_frames"] == second_frames
    g.set_channel_mask([k, c])                                 # the squelch opens channel c on THIS block, before its audio part
    g.demodulate_block(want=("pcm",))
    ev = g.ctcss_read()
    g.close()
    if expect_first == "decision":
        assert ev.shape[1] == 1                                # frames [N, 2N) complete; [2N, 2.5N) in progress
        assert ev["index"][c, 0] == c % 38 and ev["detected"][c, 0] == 1, ev[c]
    else:
        assert ev.shape[1] == 2                                # [0, N) completed by the pending part (channel c joined at N / 2), [N, 2N)
        assert ev["index"][c, 0] == -1 and ev["detected"][c, 0] == 0, ev[c]
        assert ev["index"][c, 1] == c % 38 and ev["detected"][c, 1] == 1, ev[c]
    assert np.all(ev["index"][k] == k % 38) and np.all(ev["detected"][k] == 1)
