"""ingest / egress formats (include/pmr_io.h) -- the recorded-IQ reader that stands in for readStream
(reference src/sdr_pmr446.c:789, SOAPY_SDR_CF32 src/shared.c:62) and the WAV / raw s16 writer that stands in for the
RtAudio sink (:585) and dsd_in's stdout pipe (src/dsd_in.c:172-178).  CPU tier: pure host code, checked against numpy /
scipy readers; GPU tier: the headless harness examples/pmr446_file.c end to end."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import wavfile

from sdr_pmr446_amd import chain, synth


def test_cf32_reader_delivers_blocks_like_readstream(tmp_path):
    x = synth.synth_iq(25000, 1.024e6, 16)
    p = tmp_path / "a.cf32"
    p.write_bytes(x.tobytes() + b"\x01\x02\x03")            # trailing partial sample is dropped
    r = chain.IqReader(str(p))
    got = [r.read(10000) for _ in range(4)]
    assert [len(g) for g in got] == [10000, 10000, 5000, 0]  # short only at end of stream, then 0
    assert np.array_equal(np.concatenate(got), x)
    r.close()
    with pytest.raises(chain.PmrError):
        chain.IqReader(str(tmp_path / "missing.cf32"))


def test_cs16_and_cu8_recordings_are_scaled_to_unit_range(tmp_path):
    rng = np.random.default_rng(1)
    s = rng.integers(-32768, 32768, size=2 * 777, dtype=np.int16)
    (tmp_path / "a.cs16").write_bytes(s.tobytes())
    r = chain.IqReader(str(tmp_path / "a.cs16"), chain.IQ_CS16)
    x = r.read(1000)
    r.close()
    assert len(x) == 777 and np.array_equal(x.view(np.float32), s.astype(np.float32) / np.float32(32768.0))
    b = rng.integers(0, 256, size=2 * 501, dtype=np.uint8)
    (tmp_path / "a.cu8").write_bytes(b.tobytes())
    r = chain.IqReader(str(tmp_path / "a.cu8"), chain.IQ_CU8)
    x = r.read(501)
    r.close()
    assert np.allclose(x.view(np.float32), (b.astype(np.float32) - 127.5) / 127.5, atol=1e-7)
    assert np.abs(x.view(np.float32)).max() <= 1.0


def test_reader_reads_stdin(tmp_path):
    x = synth.synth_iq(3000, 1.024e6, 16)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from sdr_pmr446_amd import chain; "
            "r = chain.IqReader('-'); a = r.read(2000); b = r.read(2000); c = r.read(10); "
            "sys.stdout.buffer.write(np.concatenate([a, b, c]).tobytes())" % os.path.dirname(os.path.dirname(__file__)))
    out = subprocess.run(["python3", "-c", code], input=x.tobytes(), capture_output=True, check=True).stdout
    assert out == x.tobytes()


@pytest.mark.parametrize("fmt,channels", [(chain.WAV_F32, 1), (chain.WAV_F32, 16), (chain.WAV_S16, 1), (chain.WAV_S16, 3)])
def test_wav_writer_round_trips_through_scipy(tmp_path, fmt, channels):
    rng = np.random.default_rng(2)
    blocks = [rng.uniform(-1, 1, size=(channels, n)).astype(np.float32) for n in (1220, 0, 1221, 7)]
    if fmt == chain.WAV_S16:
        blocks = [(b * 32767).astype(np.int16) for b in blocks]
    p = str(tmp_path / "a.wav")
    w = chain.WavWriter(p, fmt, 12500, channels)             # AUDIO_SAMPLERATE, src/sdr_pmr446.c:24
    for b in blocks:
        w.write(b)
    w.close()
    rate, data = wavfile.read(p)
    want = np.concatenate(blocks, axis=1).T
    assert rate == 12500 and data.dtype == want.dtype
    assert np.array_equal(data.reshape(want.shape[0], -1), want)


def test_raw_s16_is_the_dsd_wire_format(tmp_path):
    pcm = (np.arange(-500, 500) * 60).astype(np.int16)
    p = str(tmp_path / "a.s16")
    w = chain.WavWriter(p, chain.RAW_S16, 48000, 1)
    w.write(pcm[:300]); w.write(pcm[300:])
    w.close()
    assert open(p, "rb").read() == pcm.astype("<i2").tobytes()   # headerless s16le, src/dsd_in.c:177
    assert not chain.load().pmr_wav_writer_open(b"-", chain.WAV_F32, 12500, 1)   # RIFF needs a seekable file


@pytest.mark.gpu
def test_headless_harness_end_to_end(tmp_path):
    """examples/pmr446_file.c on a recording == the same blocks through the Python binding (same library, same split)."""
    from sdr_pmr446_amd import build
    exe = build.build_example()
    fs, M, n = 1.024e6, 16, 350000
    x = synth.synth_iq(n, fs, M)
    (tmp_path / "in.cf32").write_bytes(x.tobytes())
    subprocess.run([exe, "chan", str(tmp_path / "in.cf32"), str(tmp_path / "all.wav"), str(fs), str(M), "-1"], check=True)
    r2 = subprocess.run([exe, "chan", str(tmp_path / "in.cf32"), str(tmp_path / "ch2.wav"), str(fs), str(M), "2"], check=True,
                        capture_output=True, text=True)
    # one channel selected = the reference's mode: only it is demodulated, its CTCSS tone is logged like ctcss_execute does (:613-626)
    assert "Acquired CTCSS code: 3 (frequency: %3.2fHz)" % chain.load().pmr_ctcss_freq(2) in r2.stderr, r2.stderr[-300:]
    ch = chain.PmrChain(fs_in=fs, num_channels=M, max_block=100000)
    audio = np.concatenate([ch.process_block(x[i:i + 100000], want=("pcm", "audio"))["audio"] for i in range(0, n, 100000)], axis=1)
    rate, data = wavfile.read(str(tmp_path / "all.wav"))
    assert rate == 12500 and data.shape == audio.T.shape
    assert np.array_equal(data, audio.T)
    rate, one = wavfile.read(str(tmp_path / "ch2.wav"))
    assert np.array_equal(one, audio[2])
    # waterfall line of every block (reference :910-915), 64 characters wide: same lines as the binding gives
    r = subprocess.run([exe, "chan", str(tmp_path / "in.cf32"), str(tmp_path / "wf.wav"), str(fs), str(M), "2", "64"], check=True,
                       capture_output=True, text=True)
    lines = [l for l in r.stderr.splitlines() if l.startswith(" > ")]
    ch.reset()
    ch.spectrum_enable(64)
    want = []
    for i in range(0, n, 100000):
        ch.process_block(x[i:i + 100000], want=("pcm",))
        psd, ntr = ch.spectrum_read()
        want.append(chain.asgram_ascii(psd, 64, ntr)[0])
    assert len(lines) == len(want) == 4 and all(l[3:3 + 64] == w for l, w in zip(lines, want))
    assert any(c != " " for c in want[0])
    # dsd mode: s16le 48 kHz stream
    xd = synth.synth_iq(450000, fs, 1)
    (tmp_path / "d.cf32").write_bytes(xd.tobytes())
    subprocess.run([exe, "dsd", str(tmp_path / "d.cf32"), str(tmp_path / "d.s16")], check=True)
    d = chain.PmrDsd()
    pcm = np.concatenate([d.process_block(xd[i:i + 200000])["pcm"] for i in range(0, 450000, 200000)])
    assert np.array_equal(np.fromfile(str(tmp_path / "d.s16"), dtype="<i2"), pcm)


@pytest.mark.gpu
def test_scan_mode_follows_the_squelch(tmp_path):
    """examples/pmr446_file.c scan = the reference's loop end to end (squelch state machine :828-874 on the GPU's RSSI, only the
    open channel demodulated :876-877, reset on detune :866-867): noise, then a carrier on channel 5, then noise again."""
    from sdr_pmr446_amd import build
    exe = build.build_example()
    fs, M, nb = 1.024e6, 16, 100000
    quiet = lambda n, sid: synth.synth_iq(n, fs, M, stream_id=sid, channels=[])
    x = np.concatenate([quiet(3 * nb, 1), synth.synth_iq(5 * nb, fs, M, stream_id=2, channels=[5], dev_hz=1500.0), quiet(3 * nb, 3)])
    (tmp_path / "in.cf32").write_bytes(x.tobytes())
    r = subprocess.run([exe, "scan", str(tmp_path / "in.cf32"), str(tmp_path / "scan.wav"), str(fs), str(M)], check=True,
                       capture_output=True, text=True)
    log = r.stderr
    assert "block 3: tuned to channel 6" in log, log[-400:]            # first block that carries the signal
    assert "block 8: left channel 6" in log, log[-400:]                # first quiet block after it
    rate, data = wavfile.read(str(tmp_path / "scan.wav"))
    # decision and demodulation within one block, like the reference: exactly the blocks that carry the signal are written
    ch = chain.PmrChain(fs_in=fs, num_channels=M, max_block=nb)
    frames = [ch.process_block(x[i:i + nb], want=("pcm",))["n_frames"] for i in range(0, len(x), nb)]
    assert rate == 12500 and len(data) == sum(frames[3:8])
    assert np.abs(data[2000:]).max() > 0.05                              # audible audio while the carrier is there


@pytest.mark.gpu
@pytest.mark.parametrize("threads,blocks", [(2, 12), (3, 9)])
def test_one_pthread_per_handle_equals_the_streams_run_serially(threads, blocks):
    """examples/pmr446_threads.c: the threading model include/pmr_chain.h promises ("one thread per handle; handles are
    independent" -- the reference is a threaded C program, src/sdr_pmr446.c:520-544, 788-931), from C.  N pthreads, each with its
    own handle on the same device, ragged blocks (0, 1, 7 samples among them), host calls and un-synchronised device calls, one
    stream with the CTCSS detector, channel-mask changes and channel resets between blocks: PCM and CTCSS events byte for byte
    what the same streams give when run one after the other.  Runs under the poison mode (inherited through the environment)."""
    from sdr_pmr446_amd import build
    build.build_example()
    r = subprocess.run([build.EXAMPLE_THREADS, str(threads), str(blocks)], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("stream ")]
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    assert len(lines) == threads and all(l.endswith(": identical") for l in lines), r.stdout
    assert "CTCSS + mask changes" in lines[1] and " 0 CTCSS events" not in lines[1], lines[1]
