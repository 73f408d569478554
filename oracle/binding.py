"""ctypes binding of the CPU oracle (oracle/liboracle_pmr.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "liboracle_pmr.so")
# PMR_ORACLE_LIB: bench.py's cpu_baseline workers load the -march=native build made on the box they run on
_LIB_PATH = os.environ.get("PMR_ORACLE_LIB") or _DEFAULT_LIB


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orc_dsp.c", "orc_chain.c", "orc_dsd.c", "orc_dsp.h", "orc_chain.h", "orc_dsd.h")]
    if _LIB_PATH != _DEFAULT_LIB:
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class OrcCfg(C.Structure):
    _fields_ = [
        ("fs_in", C.c_double), ("num_channels", C.c_uint), ("channel_width_hz", C.c_double),
        ("dcblock_alpha", C.c_float), ("resamp_As", C.c_float), ("pfb_m", C.c_uint), ("pfb_As", C.c_float),
        ("fm_kf", C.c_float), ("audio_gain", C.c_float), ("lowpass", C.c_int), ("deemph_fir", C.c_int),
        ("max_block", C.c_uint), ("only_channel", C.c_int), ("ctcss_block", C.c_uint),
        ("hp_taps", C.POINTER(C.c_float)), ("hp_len", C.c_uint),
        ("lp_taps", C.POINTER(C.c_float)), ("lp_len", C.c_uint),
        ("deemph_taps", C.POINTER(C.c_float)), ("deemph_len", C.c_uint),
    ]


class OrcTaps(C.Structure):
    _fields_ = [
        ("resampled", C.c_void_p), ("resampled_cap", C.c_uint), ("n_resampled", C.c_uint),
        ("fm", C.c_void_p), ("ctcss_lp", C.c_void_p), ("audio", C.c_void_p), ("stride", C.c_uint),
        ("ctcss_events", C.c_void_p), ("ctcss_cap", C.c_uint), ("ctcss_n", C.c_uint),
    ]


class OrcDsdCfg(C.Structure):
    _fields_ = [("fs_in", C.c_double), ("sig_rate", C.c_double), ("audio_rate", C.c_double),
                ("dcblock_alpha", C.c_float), ("resamp_As", C.c_float), ("fm_kf", C.c_float), ("max_block", C.c_uint)]


CTCSS_EVENT = np.dtype([("index", np.int32), ("detected", np.int32), ("max_power", np.float32), ("avg_power", np.float32)])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_chain_default_cfg.argtypes = [C.POINTER(OrcCfg)]
        L.orc_chain_create.argtypes = [C.POINTER(OrcCfg)]
        L.orc_chain_create.restype = C.c_void_p
        L.orc_chain_reset.argtypes = [C.c_void_p]
        L.orc_chain_seek.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_chain_seek.restype = C.c_int
        L.orc_chain_destroy.argtypes = [C.c_void_p]
        L.orc_chain_reset_channel.argtypes = [C.c_void_p, C.c_uint]
        L.orc_chain_max_frames.argtypes = [C.c_void_p]
        L.orc_chain_max_frames.restype = C.c_uint
        L.orc_chain_max_resampled.argtypes = [C.c_void_p]
        L.orc_chain_max_resampled.restype = C.c_uint
        L.orc_chain_process_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint,
                                              C.POINTER(C.c_uint), C.c_void_p, C.c_void_p, C.POINTER(OrcTaps)]
        L.orc_chain_process_block.restype = C.c_int
        L.orc_firdes_kaiser.argtypes = [C.c_uint, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_kaiser_beta_As.argtypes = [C.c_float]
        L.orc_kaiser_beta_As.restype = C.c_float
        L.orc_estimate_req_filter_len.argtypes = [C.c_float, C.c_float]
        L.orc_estimate_req_filter_len.restype = C.c_uint
        L.orc_nco_constrain.argtypes = [C.c_float]
        L.orc_nco_constrain.restype = C.c_uint32
        L.orc_pcm_from_float.argtypes = [C.c_float]
        L.orc_pcm_from_float.restype = C.c_int16
        L.orc_msresamp_crcf_create.argtypes = [C.c_float, C.c_float]
        L.orc_msresamp_crcf_create.restype = C.c_void_p
        L.orc_msresamp_crcf_destroy.argtypes = [C.c_void_p]
        L.orc_msresamp_crcf_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.POINTER(C.c_uint)]
        L.orc_firpfbch_crcf_create_kaiser.argtypes = [C.c_uint, C.c_uint, C.c_float]
        L.orc_firpfbch_crcf_create_kaiser.restype = C.c_void_p
        L.orc_firpfbch_crcf_destroy.argtypes = [C.c_void_p]
        L.orc_firpfbch_crcf_analyzer_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_iirfilt_crcf_create_dc_blocker.argtypes = [C.c_float]
        L.orc_iirfilt_crcf_create_dc_blocker.restype = C.c_void_p
        L.orc_iirfilt_crcf_destroy.argtypes = [C.c_void_p]
        L.orc_iirfilt_crcf_execute_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        L.orc_chain_info.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        L.orc_chain_info.restype = C.c_uint
        L.orc_chain_design.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_uint]
        L.orc_chain_design.restype = C.c_uint
        L.orc_fft_create.argtypes = [C.c_uint]
        L.orc_fft_create.restype = C.c_void_p
        L.orc_fft_destroy.argtypes = [C.c_void_p]
        L.orc_fft_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dsd_default_cfg.argtypes = [C.POINTER(OrcDsdCfg)]
        L.orc_dsd_create.argtypes = [C.POINTER(OrcDsdCfg)]
        L.orc_dsd_create.restype = C.c_void_p
        L.orc_dsd_reset.argtypes = [C.c_void_p]
        L.orc_dsd_destroy.argtypes = [C.c_void_p]
        L.orc_dsd_max_out.argtypes = [C.c_void_p]
        L.orc_dsd_max_out.restype = C.c_uint
        L.orc_dsd_max_resampled.argtypes = [C.c_void_p]
        L.orc_dsd_max_resampled.restype = C.c_uint
        L.orc_dsd_info.argtypes = [C.c_void_p, C.c_int]
        L.orc_dsd_info.restype = C.c_uint
        L.orc_dsd_design.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint]
        L.orc_dsd_design.restype = C.c_uint
        L.orc_dsd_process_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint,
                                            C.POINTER(C.c_uint), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
        L.orc_msresamp_rrrf_create.argtypes = [C.c_float, C.c_float]
        L.orc_msresamp_rrrf_create.restype = C.c_void_p
        L.orc_msresamp_rrrf_destroy.argtypes = [C.c_void_p]
        L.orc_msresamp_rrrf_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.POINTER(C.c_uint)]
        L.orc_asgramcf_create.argtypes = [C.c_uint]
        L.orc_asgramcf_create.restype = C.c_void_p
        L.orc_asgramcf_destroy.argtypes = [C.c_void_p]
        L.orc_asgramcf_set_scale.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.orc_asgramcf_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.orc_asgramcf_execute.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
        _lib = L
    return _lib


def firdes_kaiser(n, fc, As, mu=0.0):
    h = np.zeros(n, dtype=np.float32)
    lib().orc_firdes_kaiser(n, fc, As, mu, h.ctypes.data)
    return h


class OracleChain:
    """Mirror of the reference loop body; see oracle/orc_chain.h."""

    def __init__(self, fs_in=1024000.0, num_channels=16, max_block=100000, audio_gain=4.0, lowpass=False,
                 deemph_fir=False, only_channel=-1, channel_width_hz=12500.0, pfb_m=13, pfb_As=80.0,
                 resamp_As=60.0, dcblock_alpha=0.0005, fm_kf=0.5):
        L = lib()
        cfg = OrcCfg()
        L.orc_chain_default_cfg(C.byref(cfg))
        cfg.fs_in = fs_in
        cfg.num_channels = num_channels
        cfg.channel_width_hz = channel_width_hz
        cfg.max_block = max_block
        cfg.audio_gain = audio_gain
        cfg.lowpass = int(lowpass)
        cfg.deemph_fir = int(deemph_fir)
        cfg.only_channel = only_channel
        cfg.pfb_m = pfb_m
        cfg.pfb_As = pfb_As
        cfg.resamp_As = resamp_As
        cfg.dcblock_alpha = dcblock_alpha
        cfg.fm_kf = fm_kf
        self.cfg = cfg
        self.M = num_channels
        self.h = L.orc_chain_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError("orc_chain_create failed")
        self.max_frames = L.orc_chain_max_frames(self.h)
        self.max_resampled = L.orc_chain_max_resampled(self.h)

    def reset(self):
        lib().orc_chain_reset(self.h)

    def seek(self, n_raw):
        """The state after n_raw zero samples, without running them (orc_chain_seek)."""
        rc = lib().orc_chain_seek(self.h, int(n_raw))
        if rc:
            raise RuntimeError("orc_chain_seek rc=%d" % rc)

    def reset_channel(self, k):
        lib().orc_chain_reset_channel(self.h, k)

    def info(self, what, idx=0):
        return lib().orc_chain_info(self.h, what, idx)

    def design(self, what, idx=0):
        n = lib().orc_chain_design(self.h, what, idx, None, 0)
        out = np.zeros(n, dtype=np.float32)
        lib().orc_chain_design(self.h, what, idx, out.ctypes.data, n)
        return out

    def design_dict(self):
        """All designed coefficients / integers of the chain (for the closed-form model in tests)."""
        h = self.info(0)
        return {
            "num_stages": h,
            "m_stage": [self.info(1, i) for i in range(h)],
            "hb": [self.design(0, i) for i in range(h)],
            "arb_step": self.info(2), "nco_dtheta": self.info(3), "arb_npfb": self.info(4), "arb_m": self.info(5),
            "arb": self.design(1), "pfb": self.design(2), "pfb_p": self.info(6),
        }

    def close(self):
        if self.h:
            lib().orc_chain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_block(self, iq, want=("pcm",)):
        """iq: complex64 array.  Returns dict with n_frames and the requested outputs, trimmed to n_frames."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        M, S = self.M, self.max_frames
        want = set(want)
        pcm = np.zeros((M, S), dtype=np.int16)
        chan = np.zeros((M, S), dtype=np.complex64) if "chan" in want else None
        rssi = np.zeros(M, dtype=np.float32) if "rssi" in want else None
        taps = OrcTaps()
        bufs = {}
        if "resampled" in want:
            bufs["resampled"] = np.zeros(self.max_resampled, dtype=np.complex64)
            taps.resampled = bufs["resampled"].ctypes.data
            taps.resampled_cap = self.max_resampled
        for name in ("fm", "ctcss_lp", "audio"):
            if name in want:
                bufs[name] = np.zeros((M, S), dtype=np.float32)
                setattr(taps, name, bufs[name].ctypes.data)
        taps.stride = S
        ev = None
        if "ctcss" in want:
            cap = S // 2441 + 2
            ev = np.zeros((M, cap), dtype=CTCSS_EVENT)
            taps.ctcss_events = ev.ctypes.data
            taps.ctcss_cap = cap
        ns = C.c_uint(0)
        rc = lib().orc_chain_process_block(
            self.h, iq.ctypes.data, len(iq), pcm.ctypes.data, S, C.byref(ns),
            chan.ctypes.data if chan is not None else None,
            rssi.ctypes.data if rssi is not None else None, C.byref(taps))
        if rc != 0:
            raise RuntimeError("orc_chain_process_block rc=%d" % rc)
        n = ns.value
        out = {"n_frames": n, "pcm": pcm[:, :n].copy()}
        if chan is not None:
            out["chan"] = chan[:, :n].copy()
        if rssi is not None:
            out["rssi"] = rssi
        if "resampled" in bufs:
            out["resampled"] = bufs["resampled"][:taps.n_resampled].copy()
        for name in ("fm", "ctcss_lp", "audio"):
            if name in bufs:
                out[name] = bufs[name][:, :n].copy()
        if ev is not None:
            out["ctcss"] = ev[:, :taps.ctcss_n].copy()
        return out


class OracleDsd:
    """Mirror of the `dsd_in` loop body (reference src/dsd_in.c:160-178); see oracle/orc_dsd.h."""

    def __init__(self, fs_in=1024000.0, sig_rate=12500.0, audio_rate=48000.0, max_block=200000):
        L = lib()
        cfg = OrcDsdCfg()
        L.orc_dsd_default_cfg(C.byref(cfg))
        cfg.fs_in, cfg.sig_rate, cfg.audio_rate, cfg.max_block = fs_in, sig_rate, audio_rate, max_block
        self.cfg = cfg
        self.h = L.orc_dsd_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError("orc_dsd_create failed")
        self.max_out = L.orc_dsd_max_out(self.h)
        self.max_resampled = L.orc_dsd_max_resampled(self.h)

    def info(self, what):
        return lib().orc_dsd_info(self.h, what)

    def design(self, what):
        n = lib().orc_dsd_design(self.h, what, None, 0)
        out = np.zeros(n, dtype=np.float32)
        lib().orc_dsd_design(self.h, what, out.ctypes.data, n)
        return out

    def reset(self):
        lib().orc_dsd_reset(self.h)

    def close(self):
        if self.h:
            lib().orc_dsd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_block(self, iq, want=("pcm",)):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        want = set(want)
        cap = self.max_out
        pcm = np.zeros(cap, dtype=np.int16)
        audio = np.zeros(cap, dtype=np.float32)
        res = np.zeros(self.max_resampled, dtype=np.complex64)
        fm = np.zeros(self.max_resampled, dtype=np.float32)
        nz, ny = C.c_uint(0), C.c_uint(0)
        rc = lib().orc_dsd_process_block(self.h, iq.ctypes.data, len(iq), pcm.ctypes.data, audio.ctypes.data, cap,
                                         C.byref(nz), res.ctypes.data, fm.ctypes.data, C.byref(ny))
        if rc != 0:
            raise RuntimeError("orc_dsd_process_block rc=%d" % rc)
        out = {"n_out": nz.value, "n_resampled": ny.value, "pcm": pcm[:nz.value].copy()}
        if "audio" in want:
            out["audio"] = audio[:nz.value].copy()
        if "resampled" in want:
            out["resampled"] = res[:ny.value].copy()
        if "fm" in want:
            out["fm"] = fm[:ny.value].copy()
        return out


class OracleAsgram:
    """asgramcf as the reference drives it (src/sdr_pmr446.c:474-476 create + set_scale(-40, 2); :911-912 write the block's
    resampled samples, execute); see oracle/orc_dsp.h for the restated liquid algorithm."""

    def __init__(self, nfft, ref=-40.0, div=2.0):
        self.nfft, self.nfftp = int(nfft), 4 * int(nfft)
        self.h = lib().orc_asgramcf_create(self.nfft)
        if not self.h:
            raise RuntimeError("orc_asgramcf_create failed (nfft must be a power of two >= 2)")
        lib().orc_asgramcf_set_scale(self.h, ref, div)

    def close(self):
        if self.h:
            lib().orc_asgramcf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def block(self, resampled):
        """One loop iteration: write + execute.  Returns dict(ascii, peakval, peakfreq, psd_db[4 nfft])."""
        x = np.ascontiguousarray(resampled, dtype=np.complex64)
        lib().orc_asgramcf_write(self.h, x.ctypes.data, len(x))
        buf = C.create_string_buffer(self.nfft + 1)
        pv, pf = C.c_float(0), C.c_float(0)
        psd = np.zeros(self.nfftp, dtype=np.float32)
        lib().orc_asgramcf_execute(self.h, buf, C.byref(pv), C.byref(pf), psd.ctypes.data)
        return {"ascii": buf.raw[:self.nfft].decode("ascii"), "peakval": pv.value, "peakfreq": pf.value, "psd_db": psd}
