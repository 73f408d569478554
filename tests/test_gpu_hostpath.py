"""Host-buffer entry points at the reference's call pattern (one readStream block per loop iteration,
src/sdr_pmr446.c:789-796): the synchronous call, and the asynchronous submit / collect pair with up to PIPE_DEPTH blocks in
flight -- same results, bit for bit, and +-1 LSB against the oracle."""
import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, CFG_REF, active_channels, pcm_diff
from sdr_pmr446_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,nb,nblk,pinned", [(CFG_REF, 100000, 9, True), (CFG2, 100000, 7, False), (CFG5, 1 << 22, 5, True)],
                         ids=["ref-point-pinned", "cfg2-pageable", "cfg5-pinned"])
def test_submit_collect_equals_synchronous_calls(cfg, nb, nblk, pinned):
    from sdr_pmr446_amd import chain
    fs, M = cfg
    ks = None if M <= 64 else list(range(0, M, 73))
    x = synth.synth_iq(nb * nblk, fs, M, channels=ks, dev_hz=1500.0)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=nb)
    sync = [g.process_block(x[b * nb:(b + 1) * nb], want=("pcm", "audio", "rssi")) for b in range(nblk)]
    g.reset()
    depth = g._L.pmr_chain_max_in_flight(g.h)
    assert depth >= 2
    bufs = [g.pinned_array(nb) if pinned else np.zeros(nb, np.complex64) for _ in range(depth)]
    got, sub = [], 0
    for b in range(nblk + depth):
        if b >= depth or b >= nblk:                      # the pipe is full (or drained of input): take the oldest block out
            if len(got) < nblk:
                got.append(g.collect_block())
        if b < nblk:
            buf = bufs[b % depth]
            buf[:] = x[b * nb:(b + 1) * nb]
            g.submit_block(buf, want=("pcm", "audio", "rssi"))
            sub += 1
    while len(got) < nblk:
        got.append(g.collect_block())
    assert g._L.pmr_chain_blocks_in_flight(g.h) == 0
    for a, b in zip(sync, got):
        assert a["n_frames"] == b["n_frames"]
        assert np.array_equal(a["pcm"], b["pcm"]) and np.array_equal(a["audio"], b["audio"])
        assert np.allclose(a["rssi"], b["rssi"], atol=1e-4, equal_nan=True)
    with pytest.raises(chain.PmrError):
        g.collect_block() if False else g._check(g._L.pmr_chain_collect_block(g.h, None, None, 0, None, None, None))
    # and against the oracle
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=nb)
    ref = np.concatenate([o.process_block(x[b * nb:(b + 1) * nb])["pcm"] for b in range(nblk)], axis=1)
    pcm = np.concatenate([r["pcm"] for r in got], axis=1)
    act = active_channels(M, ks, fs)
    assert pcm.shape == ref.shape and pcm_diff(pcm[act], ref[act]).max() <= 1
    g.close(); o.close()


def test_too_many_blocks_in_flight_is_refused_not_dropped():
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=50000)
    x = synth.synth_iq(50000, fs, M)
    depth = g._L.pmr_chain_max_in_flight(g.h)
    for _ in range(depth):
        g.submit_block(x)
    with pytest.raises(chain.PmrError):
        g.submit_block(x)
    assert len(g._pending) == depth                    # the refused block was never queued (chain.py appends after success)
    frames = [g.collect_block()["n_frames"] for _ in range(depth)]
    assert sum(frames) > 0 and g._L.pmr_chain_blocks_in_flight(g.h) == 0
    # the synchronous call works again afterwards, mixed with device-entry calls on the two-stream pipeline
    assert g.process_block(x)["n_frames"] > 0


@pytest.mark.parametrize("fmt", ["cs16", "cu8"])
def test_integer_ingest_formats_are_converted_on_the_device(fmt):
    """pmr_chain_submit_block_fmt: int16 / uint8 I/Q (include/pmr_io.h formats) cross PCIe as they are and are converted on the
    device with the rules of the host-side reader (pmr_io.c) -- PCM bit-identical to feeding the host-converted cf32."""
    from sdr_pmr446_amd import chain
    fs, M = CFG2
    nb, nblk = 100001, 4                                  # odd block size: the converter's tail path runs too
    x = synth.synth_iq(nb * nblk, fs, M, dev_hz=1500.0)
    xi = np.empty(2 * len(x), np.float32); xi[0::2] = x.real; xi[1::2] = x.imag
    if fmt == "cs16":
        raw = np.clip(np.round(xi * 32768.0 * 1.5), -32768, 32767).astype(np.int16)
        host = (raw.astype(np.float32) * np.float32(1.0 / 32768.0))
        code = chain.IQ_CS16
    else:
        raw = np.clip(np.round(xi * 127.5 * 1.5 + 127.5), 0, 255).astype(np.uint8)
        host = (raw.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)
        code = chain.IQ_CU8
    xc = (host[0::2] + 1j * host[1::2]).astype(np.complex64)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=nb)
    ref = [g.process_block(xc[b * nb:(b + 1) * nb])["pcm"] for b in range(nblk)]
    g.reset()
    got = []
    for b in range(nblk):
        g.submit_block(raw[2 * b * nb:2 * (b + 1) * nb], fmt=code)
        if b >= 2:
            got.append(g.collect_block()["pcm"])
    while len(got) < nblk:
        got.append(g.collect_block()["pcm"])
    for a, b in zip(ref, got):
        assert np.array_equal(a, b) and a.shape[1] > 100
    g.close()


@pytest.mark.parametrize("cfg,fmt,sizes", [(CFG_REF, "cu8", [100000, 99999, 7, 0, 100000, 65537, 3]),
                                           (CFG_REF, "cs16", [100000, 100001, 1, 99998]),
                                           (CFG2, "cu8", [250000, 1 << 18, (1 << 18) + 1, 300001]),
                                           (CFG3, "cs16", [1 << 18, 200003, 150001]),
                                           (CFG5, "cu8", [1 << 18, 200003])],
                         ids=["ref-point-cu8", "ref-point-cs16", "cfg2-cu8-around-threshold", "cfg3-cs16", "cfg5-cu8"])
def test_sync_integer_formats_read_in_place_equal_the_converted_cf32_call(cfg, fmt, sizes):
    """pmr_chain_process_block_fmt: the synchronous call on the receiver's own samples (the reference's radio is an RTL-SDR: uint8
    pairs, README.md:12; SoapySDR widens them to the cf32 of readStream, src/shared.c:62).  Blocks of up to 2^18 samples in
    pmr_host_alloc memory are read in place by the front end and converted as it loads them (2 / 4 bytes per sample on the host
    link); larger or pageable blocks go through H2D + the conversion kernel.  Either way PCM and audio are BIT-IDENTICAL to the cf32
    call on the host-converted samples -- ragged sizes, a view at an odd offset inside the pinned allocation (unaligned groups) and a
    pageable copy included."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    n = sum(sizes)
    x = synth.synth_iq(n, fs, M, dev_hz=1500.0, channels=None if M <= 16 else list(range(0, M, M // 16)))
    xi = np.empty(2 * n, np.float32); xi[0::2] = x.real; xi[1::2] = x.imag
    if fmt == "cs16":
        raw = np.clip(np.round(xi * 32768.0 * 1.5), -32768, 32767).astype(np.int16)
        host = raw.astype(np.float32) * np.float32(1.0 / 32768.0)
        code = chain.IQ_CS16
    else:
        raw = np.clip(np.round(xi * 127.5 * 1.5 + 127.5), 0, 255).astype(np.uint8)
        host = (raw.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)
        code = chain.IQ_CU8
    xc = (host[0::2] + 1j * host[1::2]).astype(np.complex64)
    mb = max(sizes)
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
    ref, pos = [], 0
    for s_ in sizes:
        ref.append(g.process_block(xc[pos:pos + s_], want=("pcm", "audio")))
        pos += s_
    for variant in ("pinned", "pinned-odd-offset", "pageable"):
        g.reset()
        pin = g.pinned_array(2 * mb + 8, raw.dtype)
        pos = 0
        for k, s_ in enumerate(sizes):
            blk = raw[2 * pos:2 * (pos + s_)]
            if variant == "pageable":
                src = blk
            else:
                off = 2 if variant == "pinned-odd-offset" else 0          # one complex sample in: groups of four no longer aligned
                src = pin[off:off + 2 * s_]
                src[:] = blk
            r = g.process_block(src, want=("pcm", "audio"), fmt=code)
            assert r["n_frames"] == ref[k]["n_frames"]
            assert np.array_equal(r["pcm"], ref[k]["pcm"]), (variant, k)
            assert np.array_equal(r["audio"], ref[k]["audio"]), (variant, k)
            pos += s_
    assert sum(r_["n_frames"] for r_ in ref) >= 5                  # (cfg5: 1024 channels, 12.5 kHz each: 460 000 samples are 5 frames)
    g.close()


@pytest.mark.parametrize("cfg,sizes", [(CFG_REF, [100000, 99999, 7, 0, 100000, 65537]), (CFG2, [250000, 1 << 18, (1 << 18) + 1, 300001]),
                                       (CFG5, [1 << 18, 200003])],
                         ids=["ref-point", "cfg2-around-threshold", "cfg5"])
def test_zero_copy_pinned_input_equals_copied_input(cfg, sizes):
    """Synchronous calls on small blocks read a pmr_host_alloc'd input in place and write the outputs straight to pinned memory
    (no copy-engine submissions); blocks above 2^18 samples, pageable inputs and PMR_ZEROCOPY=0 go through the copies.  Same
    stream fed both ways -- including a view at an odd offset inside the pinned allocation, which takes the front end's unaligned
    tile path -- must agree bit for bit (pcm, audio, chan) and to rounding in rssi."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    ks = None if M <= 64 else list(range(0, M, 73))
    x = synth.synth_iq(sum(sizes), fs, M, channels=ks, dev_hz=1500.0)
    mb = max(sizes)
    ga = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
    gb = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb)
    pin = ga.pinned_array(mb + 1)
    pos = 0
    for i, n in enumerate(sizes):
        blk = x[pos:pos + n]
        pos += n
        view = pin[i & 1:(i & 1) + n]                   # odd blocks start 8 bytes into the allocation (not 16-byte aligned)
        view[:] = blk
        a = ga.process_block(view, want=("pcm", "audio", "chan", "rssi"))
        b = gb.process_block(blk.copy(), want=("pcm", "audio", "chan", "rssi"))        # pageable: copy path
        assert a["n_frames"] == b["n_frames"]
        for k in ("pcm", "audio", "chan"):
            assert np.array_equal(a[k], b[k]), (k, i, n)
        assert np.allclose(a["rssi"], b["rssi"], atol=1e-4, equal_nan=True)
    ga.close(); gb.close()


@pytest.mark.parametrize("cfg,sizes,opts,ctcss", [(CFG_REF, [100000, 99999, 7, 0, 100000, 65537, 100000], {}, False),
                                                    (CFG_REF, [100000] * 6, {}, True),
                                                    (CFG2, [250000, 1 << 18, 300001], {"lowpass": True}, False),
                                                    (CFG5, [1 << 22, 3000001, 1 << 21], {}, False)],
                         ids=["ref-ragged", "ref-ctcss", "cfg2-lowpass", "cfg5"])
def test_two_step_form_equals_the_single_call(cfg, sizes, opts, ctcss):
    """pmr_chain_channelize_block + pmr_chain_demodulate_block (the reference's order: squelch decision between channelizer and
    demodulator, :828-877) with an unchanged mask return what pmr_chain_process_block_f32 returns, bit for bit -- also when a
    channelized block is never demodulated (its audio part is still pushed through the stateful stages by the next call)."""
    from sdr_pmr446_amd import chain
    fs, M = cfg
    ks = None if M <= 64 else list(range(0, M, 73))
    x = synth.synth_iq(sum(sizes), fs, M, channels=ks, dev_hz=1500.0)
    mb = max(sizes)
    ga = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb, **opts)
    gb = chain.PmrChain(fs_in=fs, num_channels=M, max_block=mb, **opts)
    if ctcss:
        for g in (ga, gb):
            g._check(g._L.pmr_chain_ctcss_enable(g.h, 1))
    pos = 0
    for i, n in enumerate(sizes):
        blk = x[pos:pos + n]
        pos += n
        a = ga.process_block(blk, want=("pcm", "audio", "chan", "rssi"))
        c = gb.channelize_block(blk, want=("rssi", "chan"))
        assert c["n_frames"] == a["n_frames"]
        assert np.array_equal(c["chan"], a["chan"]) and np.allclose(c["rssi"], a["rssi"], atol=1e-4, equal_nan=True)
        if i == 1:
            continue                                    # this block is never demodulated in the two-step chain
        b = gb.demodulate_block(want=("pcm", "audio"))
        assert b["n_frames"] == a["n_frames"]
        assert np.array_equal(a["pcm"], b["pcm"]) and np.array_equal(a["audio"], b["audio"]), (i, n)
        if ctcss:
            ea, eb = ga.ctcss_read(), gb.ctcss_read()
            assert ea.shape == eb.shape and ea.tobytes() == eb.tobytes(), i
    with pytest.raises(chain.PmrError):
        gb.demodulate_block()                           # nothing pending
    ga.close(); gb.close()


def test_two_step_form_opens_the_squelch_on_the_same_block():
    """The reference demodulates the block whose RSSI opened the squelch.  Noise, then a carrier on channel 5: with the two-step
    form the first block that carries the signal already yields its audio, +-1 LSB of the oracle demodulating channel 5."""
    from sdr_pmr446_amd import chain
    fs, M, nb = CFG_REF[0], CFG_REF[1], 100000
    x = np.concatenate([synth.synth_iq(2 * nb, fs, M, stream_id=1, channels=[]),
                        synth.synth_iq(3 * nb, fs, M, stream_id=2, channels=[5], dev_hz=1500.0)])
    g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=nb)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=nb, only_channel=5)
    L = g._L
    import ctypes as C
    st = chain.Squelch()
    L.pmr_squelch_init(C.byref(st))
    g.set_channel_mask([])
    opened_at = None
    for b in range(5):
        blk = x[b * nb:(b + 1) * nb]
        ref = o.process_block(blk, want=("pcm",))["pcm"]
        c = g.channelize_block(blk, want=("rssi",))
        if L.pmr_squelch_update(C.byref(st), c["rssi"].ctypes.data, M, None, 0, 18.0, 0):
            g.set_channel_mask([st.active_chan] if st.state == 1 else [])
            if st.state == 1 and opened_at is None:
                opened_at = b
        d = g.demodulate_block(want=("pcm",))
        if st.state == 1:
            assert st.active_chan == 5
            got, want = d["pcm"][5].astype(np.int32), ref[5].astype(np.int32)
            skip = 700 if b == opened_at else 0         # the carrier starts inside the filters' history
            assert np.abs(got[skip:] - want[skip:]).max() <= 1
            assert np.abs(got).max() > 1000
    assert opened_at == 2
    g.close(); o.close()


@pytest.mark.parametrize("mask", [None, [5], [0, 3, 15]], ids=["all-open", "one-open", "three-open"])
def test_rssi_finish_riding_in_the_fir_launch_equals_its_own_kernel(mask):
    """Synchronous small-block calls let the RSSI finish ride in the audio FIR's launch (one extra workgroup of k_fir_mfma4, a
    kernel boundary less); the two-step form and blocks of many tiles launch k_rssi_finish.  Same sums in the same order: the
    figures must be EQUAL, with every channel open (2-D grid + an extra row) and with a channel mask (1-D grid + an extra
    workgroup)."""
    from sdr_pmr446_amd import chain
    fs, M = CFG_REF
    sizes = [100000, 99999, 4000, 100000]
    x = synth.synth_iq(sum(sizes), fs, M, dev_hz=1500.0)
    ga = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
    gb = chain.PmrChain(fs_in=fs, num_channels=M, max_block=max(sizes))
    for g in (ga, gb):
        g.set_channel_mask(mask)
    pos = 0
    for n in sizes:
        blk = x[pos:pos + n]
        pos += n
        a = ga.process_block(blk, want=("pcm", "rssi"))                      # rider
        c = gb.channelize_block(blk, want=("rssi",))                          # k_rssi_finish
        b = gb.demodulate_block(want=("pcm",))
        assert a["rssi"].tobytes() == c["rssi"].tobytes(), n
        assert np.array_equal(a["pcm"], b["pcm"])
    ga.close(); gb.close()
