"""Host-side mirror of include/pmr_chain.h (ctypes over libpmr446_hip.so).

This is plumbing: it owns no arithmetic.  Every call goes to the C-ABI, which enqueues the gfx950 kernels.
If the library (or a HIP device) is missing the import/`PmrChain()` raises -- there is no CPU path here.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

PMR_OK, PMR_EINVAL, PMR_ERANGE, PMR_EHIP, PMR_ENOMEM = 0, 1, 2, 3, 4

INFO_NUM_STAGES, INFO_M_STAGE, INFO_ARB_STEP, INFO_NCO_DTHETA, INFO_ARB_NPFB, INFO_ARB_M, INFO_PFB_P, INFO_CARRY_AT_LOAD = range(8)
DESIGN_HALFBAND, DESIGN_ARB, DESIGN_PFB, DESIGN_DEEMPH = range(4)
DEBUG_RESAMPLED, DEBUG_FM, DEBUG_CTCSS_LP = range(3)

#: every symbol include/pmr_chain.h declares
ABI_SYMBOLS = [
    "pmr_chain_default_cfg", "pmr_chain_create", "pmr_chain_create_error", "pmr_chain_seek", "pmr_chain_position", "pmr_chain_reset", "pmr_chain_destroy", "pmr_chain_max_frames",
    "pmr_chain_num_channels", "pmr_chain_last_error", "pmr_chain_process_block", "pmr_chain_process_block_f32", "pmr_chain_process_block_fmt",
    "pmr_chain_process_block_device", "pmr_chain_synchronize", "pmr_chain_set_overlap", "pmr_chain_stream", "pmr_chain_profile_enable",
    "pmr_chain_profile_reset", "pmr_chain_profile_count", "pmr_chain_profile_name", "pmr_chain_profile_get",
    "pmr_chain_info", "pmr_chain_design", "pmr_chain_debug_enable", "pmr_chain_debug_read", "pmr_debug_poison", "pmr_debug_lds_probe",
    "pmr_cfg_info", "pmr_cfg_design", "pmr_cfg_max_frames", "pmr_cfg_plan_block",
    "pmr_squelch_init", "pmr_find_max_rssi_channel", "pmr_squelch_update",
    "pmr_chain_spectrum_enable", "pmr_chain_spectrum_read", "pmr_asgram_ascii",
    "pmr_chain_channelize_block", "pmr_chain_demodulate_block",
    "pmr_chain_ctcss_enable", "pmr_chain_ctcss_read", "pmr_ctcss_freq", "pmr_chain_set_channel_mask", "pmr_chain_reset_channel",
    "pmr_chain_submit_block", "pmr_chain_submit_block_fmt", "pmr_chain_collect_block", "pmr_chain_blocks_in_flight", "pmr_chain_max_in_flight",
    "pmr_host_alloc", "pmr_host_free", "pmr_chain_wait_input_event", "pmr_chain_synchronize_input",
    # include/pmr_mem.h
    "pmr_device_alloc", "pmr_device_free", "pmr_memcpy_h2d", "pmr_memcpy_d2h", "pmr_device_synchronize",
    "pmr_synth_default_cfg", "pmr_synth_iq_device",
    # include/pmr_dsd.h
    "pmr_dsd_default_cfg", "pmr_dsd_create", "pmr_dsd_reset", "pmr_dsd_destroy", "pmr_dsd_max_out",
    "pmr_dsd_last_error", "pmr_dsd_process_block", "pmr_dsd_process_block_device", "pmr_dsd_synchronize",
    "pmr_dsd_debug_read", "pmr_dsd_plan_block", "pmr_dsd_cfg_info",
    # include/pmr_io.h
    "pmr_iq_reader_open", "pmr_iq_reader_read", "pmr_iq_reader_close",
    "pmr_wav_writer_open", "pmr_wav_writer_write_f32", "pmr_wav_writer_write_s16", "pmr_wav_writer_close",
]

IQ_CF32, IQ_CS16, IQ_CU8 = 0, 1, 2      # include/pmr_io.h ingest formats

CTCSS_EVENT = np.dtype([("index", np.int32), ("detected", np.int32), ("max_power", np.float32),
                        ("avg_power", np.float32)])


class Squelch(C.Structure):
    """pmr_squelch: state (0 scanning / 1 tuned), active_chan, rssi -- SURVEY s8 row f1."""
    _fields_ = [("state", C.c_int), ("active_chan", C.c_int), ("rssi", C.c_float)]


class PlanState(C.Structure):
    _fields_ = [("n_raw", C.c_uint64), ("arb_phase", C.c_uint32), ("leftover", C.c_uint)]


class DsdCfg(C.Structure):
    """pmr_dsd_cfg (include/pmr_dsd.h)."""
    _fields_ = [("fs_in", C.c_double), ("sig_rate", C.c_double), ("audio_rate", C.c_double),
                ("dcblock_alpha", C.c_float), ("resamp_As", C.c_float), ("fm_kf", C.c_float),
                ("max_block", C.c_uint), ("device", C.c_int)]


class DsdPlanState(C.Structure):
    _fields_ = [("n_raw", C.c_uint64), ("n_resampled", C.c_uint64), ("down_phase", C.c_uint32)]


class SynthCfg(C.Structure):
    """pmr_synth_cfg (include/pmr_mem.h)."""
    _fields_ = [("fs_in", C.c_double), ("num_channels", C.c_uint), ("stream_id", C.c_uint), ("snr_db", C.c_double),
                ("dev_hz", C.c_double), ("ctcss_dev_hz", C.c_double), ("period_log2", C.c_uint), ("channel_step", C.c_uint)]


class PmrCfg(C.Structure):
    _fields_ = [
        ("fs_in", C.c_double), ("num_channels", C.c_uint), ("channel_width_hz", C.c_double),
        ("dcblock_alpha", C.c_float), ("resamp_As", C.c_float), ("pfb_m", C.c_uint), ("pfb_As", C.c_float),
        ("fm_kf", C.c_float), ("audio_gain", C.c_float), ("lowpass", C.c_int), ("deemph_fir", C.c_int),
        ("max_block", C.c_uint), ("device", C.c_int),
        ("hp_taps", C.POINTER(C.c_float)), ("hp_len", C.c_uint),
        ("lp_taps", C.POINTER(C.c_float)), ("lp_len", C.c_uint),
        ("deemph_taps", C.POINTER(C.c_float)), ("deemph_len", C.c_uint),
    ]


_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load libpmr446_hip.so (building it in-tree first if needed) and declare the C-ABI prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # Two HIP runtimes can end up in one process: this library links the system libamdhip64.so.7, a PyTorch
    # wheel bundles its own (different SONAME).  They coexist as long as PyTorch's is initialised FIRST, so when
    # torch is installed let it initialise before libpmr446_hip.so pulls in the system runtime.
    if not os.environ.get("PMR_NO_TORCH"):           # (lean tools that never touch torch skip its multi-second import)
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    path = _build.LIB
    if build_if_missing:
        path = _build.build()
    alt = os.environ.get("PMR_LIBRARY")          # another BUILD of this library (same ABI): A/B of two builds on one GPU box
    if alt:
        path = alt
    if not os.path.exists(path):
        raise RuntimeError("libpmr446_hip.so is missing: run sdr_pmr446_amd/build.py (no CPU fallback exists)")
    L = C.CDLL(path)
    vp, u, i = C.c_void_p, C.c_uint, C.c_int
    L.pmr_chain_default_cfg.argtypes = [C.POINTER(PmrCfg)]
    L.pmr_chain_default_cfg.restype = None
    L.pmr_chain_create.argtypes = [C.POINTER(PmrCfg)]
    L.pmr_chain_create.restype = vp
    for name in ("pmr_chain_reset", "pmr_chain_destroy", "pmr_chain_synchronize", "pmr_chain_profile_reset"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i
    for name in ("pmr_chain_max_frames", "pmr_chain_num_channels", "pmr_chain_profile_count"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = u
    L.pmr_chain_last_error.argtypes = [vp]
    L.pmr_chain_last_error.restype = C.c_char_p
    L.pmr_chain_create_error.argtypes = []
    L.pmr_chain_create_error.restype = C.c_char_p
    L.pmr_chain_seek.argtypes = [vp, C.c_uint64]
    L.pmr_chain_seek.restype = i
    L.pmr_chain_position.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pmr_chain_position.restype = None
    L.pmr_chain_stream.argtypes = [vp]
    L.pmr_chain_stream.restype = vp
    L.pmr_chain_process_block.argtypes = [vp, vp, u, vp, u, C.POINTER(u), vp, vp]
    L.pmr_chain_process_block.restype = i
    L.pmr_chain_process_block_f32.argtypes = [vp, vp, u, vp, vp, u, C.POINTER(u), vp, vp]
    L.pmr_chain_process_block_f32.restype = i
    L.pmr_chain_process_block_fmt.argtypes = [vp, vp, i, u, vp, vp, u, C.POINTER(u), vp, vp]
    L.pmr_chain_process_block_fmt.restype = i
    L.pmr_chain_process_block_device.argtypes = [vp, vp, u, vp, vp, u, C.POINTER(u), vp, vp]
    L.pmr_chain_process_block_device.restype = i
    L.pmr_chain_set_overlap.argtypes = [vp, i]
    L.pmr_chain_set_overlap.restype = i
    L.pmr_chain_profile_enable.argtypes = [vp, i]
    L.pmr_chain_profile_enable.restype = i
    L.pmr_chain_profile_name.argtypes = [vp, u]
    L.pmr_chain_profile_name.restype = C.c_char_p
    L.pmr_chain_profile_get.argtypes = [vp, u, C.POINTER(C.c_double), C.POINTER(u)]
    L.pmr_chain_profile_get.restype = i
    L.pmr_chain_info.argtypes = [vp, i, u]
    L.pmr_chain_info.restype = u
    L.pmr_chain_design.argtypes = [vp, i, u, vp, u]
    L.pmr_chain_design.restype = u
    L.pmr_chain_debug_enable.argtypes = [vp, i]
    L.pmr_chain_debug_enable.restype = i
    L.pmr_debug_poison.argtypes = [i]
    L.pmr_debug_poison.restype = i
    L.pmr_debug_lds_probe.argtypes = [vp, u, u]
    L.pmr_debug_lds_probe.restype = i
    L.pmr_chain_debug_read.argtypes = [vp, i, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.pmr_chain_debug_read.restype = i
    L.pmr_chain_submit_block.argtypes = [vp, vp, u, u]
    L.pmr_chain_submit_block.restype = i
    L.pmr_chain_submit_block_fmt.argtypes = [vp, vp, i, u, u]
    L.pmr_chain_submit_block_fmt.restype = i
    L.pmr_chain_collect_block.argtypes = [vp, vp, vp, u, C.POINTER(u), vp, vp]
    L.pmr_chain_collect_block.restype = i
    for name in ("pmr_chain_blocks_in_flight", "pmr_chain_max_in_flight"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = u
    L.pmr_host_alloc.argtypes = [C.c_size_t]
    L.pmr_host_alloc.restype = vp
    L.pmr_host_free.argtypes = [vp]
    L.pmr_host_free.restype = None
    L.pmr_chain_wait_input_event.argtypes = [vp, vp]
    L.pmr_chain_wait_input_event.restype = i
    L.pmr_chain_synchronize_input.argtypes = [vp]
    L.pmr_chain_synchronize_input.restype = i
    L.pmr_device_alloc.argtypes = [C.c_size_t, i]
    L.pmr_device_alloc.restype = vp
    L.pmr_device_free.argtypes = [vp]
    L.pmr_device_free.restype = None
    L.pmr_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.pmr_memcpy_h2d.restype = i
    L.pmr_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.pmr_memcpy_d2h.restype = i
    L.pmr_device_synchronize.argtypes = []
    L.pmr_device_synchronize.restype = i
    L.pmr_synth_default_cfg.argtypes = [C.POINTER(SynthCfg), C.c_double, u]
    L.pmr_synth_default_cfg.restype = None
    L.pmr_synth_iq_device.argtypes = [C.POINTER(SynthCfg), vp, C.c_uint64, C.c_size_t]
    L.pmr_synth_iq_device.restype = i
    L.pmr_chain_set_channel_mask.argtypes = [vp, vp, u]
    L.pmr_chain_set_channel_mask.restype = i
    L.pmr_chain_reset_channel.argtypes = [vp, u]
    L.pmr_chain_reset_channel.restype = i
    L.pmr_chain_ctcss_enable.argtypes = [vp, i]
    L.pmr_chain_ctcss_enable.restype = i
    L.pmr_chain_ctcss_read.argtypes = [vp, vp, u, C.POINTER(u)]
    L.pmr_chain_ctcss_read.restype = i
    L.pmr_chain_channelize_block.argtypes = [vp, vp, u, C.POINTER(u), vp, u, vp]
    L.pmr_chain_channelize_block.restype = i
    L.pmr_chain_demodulate_block.argtypes = [vp, vp, vp, u, C.POINTER(u)]
    L.pmr_chain_demodulate_block.restype = i
    L.pmr_ctcss_freq.argtypes = [i]
    L.pmr_ctcss_freq.restype = C.c_float
    L.pmr_chain_spectrum_enable.argtypes = [vp, u]
    L.pmr_chain_spectrum_enable.restype = i
    L.pmr_chain_spectrum_read.argtypes = [vp, vp, u, C.POINTER(u)]
    L.pmr_chain_spectrum_read.restype = i
    L.pmr_asgram_ascii.argtypes = [vp, u, u, C.c_float, C.c_float, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pmr_asgram_ascii.restype = i
    L.pmr_squelch_init.argtypes = [C.POINTER(Squelch)]
    L.pmr_squelch_init.restype = None
    L.pmr_find_max_rssi_channel.argtypes = [vp, u, vp, u, C.POINTER(C.c_float)]
    L.pmr_find_max_rssi_channel.restype = i
    L.pmr_squelch_update.argtypes = [C.POINTER(Squelch), vp, u, vp, u, C.c_float, i]
    L.pmr_squelch_update.restype = i
    L.pmr_cfg_info.argtypes = [C.POINTER(PmrCfg), i, u]
    L.pmr_cfg_info.restype = u
    L.pmr_cfg_design.argtypes = [C.POINTER(PmrCfg), i, u, vp, u]
    L.pmr_cfg_design.restype = u
    L.pmr_cfg_max_frames.argtypes = [C.POINTER(PmrCfg)]
    L.pmr_cfg_max_frames.restype = u
    L.pmr_cfg_plan_block.argtypes = [C.POINTER(PmrCfg), C.POINTER(PlanState), u, C.POINTER(u), C.POINTER(u)]
    L.pmr_cfg_plan_block.restype = i
    L.pmr_iq_reader_open.argtypes = [C.c_char_p, i]
    L.pmr_iq_reader_open.restype = vp
    L.pmr_iq_reader_read.argtypes = [vp, vp, u]
    L.pmr_iq_reader_read.restype = i
    L.pmr_iq_reader_close.argtypes = [vp]
    L.pmr_iq_reader_close.restype = i
    L.pmr_wav_writer_open.argtypes = [C.c_char_p, i, u, u]
    L.pmr_wav_writer_open.restype = vp
    L.pmr_wav_writer_write_f32.argtypes = [vp, vp, u, u]
    L.pmr_wav_writer_write_f32.restype = i
    L.pmr_wav_writer_write_s16.argtypes = [vp, vp, u, u]
    L.pmr_wav_writer_write_s16.restype = i
    L.pmr_wav_writer_close.argtypes = [vp]
    L.pmr_wav_writer_close.restype = i
    L.pmr_dsd_default_cfg.argtypes = [C.POINTER(DsdCfg)]
    L.pmr_dsd_default_cfg.restype = None
    L.pmr_dsd_create.argtypes = [C.POINTER(DsdCfg)]
    L.pmr_dsd_create.restype = vp
    for name in ("pmr_dsd_reset", "pmr_dsd_destroy", "pmr_dsd_synchronize"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i
    L.pmr_dsd_max_out.argtypes = [vp]
    L.pmr_dsd_max_out.restype = u
    L.pmr_dsd_last_error.argtypes = [vp]
    L.pmr_dsd_last_error.restype = C.c_char_p
    L.pmr_dsd_process_block.argtypes = [vp, vp, u, vp, vp, u, C.POINTER(u)]
    L.pmr_dsd_process_block.restype = i
    L.pmr_dsd_process_block_device.argtypes = [vp, vp, u, vp, vp, u, C.POINTER(u)]
    L.pmr_dsd_process_block_device.restype = i
    L.pmr_dsd_debug_read.argtypes = [vp, i, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.pmr_dsd_debug_read.restype = i
    L.pmr_dsd_plan_block.argtypes = [C.POINTER(DsdCfg), C.POINTER(DsdPlanState), u, C.POINTER(u), C.POINTER(u)]
    L.pmr_dsd_plan_block.restype = i
    L.pmr_dsd_cfg_info.argtypes = [C.POINTER(DsdCfg), i]
    L.pmr_dsd_cfg_info.restype = u
    _lib = L
    return L


class DeviceBuffer:
    """HBM through the library's own HIP runtime (include/pmr_mem.h): no PyTorch needed to hold a block on the device."""

    def __init__(self, nbytes, device=-1):
        self._L = load()
        self.nbytes = int(nbytes)
        self.ptr = self._L.pmr_device_alloc(self.nbytes, device)
        if not self.ptr:
            raise PmrError("pmr_device_alloc(%d) failed" % nbytes)

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        if self._L.pmr_memcpy_h2d(self.ptr + offset, arr.ctypes.data, arr.nbytes):
            raise PmrError("pmr_memcpy_h2d failed")

    def download(self, dtype, count, offset=0):
        out = np.empty(int(count), dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        if self._L.pmr_memcpy_d2h(out.ctypes.data, self.ptr + offset, out.nbytes):
            raise PmrError("pmr_memcpy_d2h failed")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self._L.pmr_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synth_iq_device(n, fs_in, num_channels, stream_id=0, n0=0, dev_hz=2500.0, ctcss_dev_hz=300.0, snr_db=30.0, period_log2=0,
                    channel_step=1, device=-1):
    """DeviceBuffer holding n cf32 samples [n0, n0 + n) of the synthetic SURVEY s8(d) stream, generated by a kernel in HBM."""
    L = load()
    buf = DeviceBuffer(int(n) * 8, device)
    cfg = SynthCfg()
    L.pmr_synth_default_cfg(C.byref(cfg), fs_in, num_channels)
    cfg.stream_id, cfg.snr_db, cfg.dev_hz, cfg.ctcss_dev_hz = stream_id, snr_db, dev_hz, ctcss_dev_hz
    cfg.period_log2, cfg.channel_step = period_log2, channel_step
    rc = L.pmr_synth_iq_device(C.byref(cfg), buf.ptr, n0, int(n))
    if rc:
        raise PmrError("pmr_synth_iq_device rc=%d" % rc)
    return buf


def device_synchronize():
    if load().pmr_device_synchronize():
        raise PmrError("pmr_device_synchronize failed")


def make_cfg(fs_in=1024000.0, num_channels=16, max_block=100000, **kw):
    """Fill a pmr_chain_cfg starting from the reference's operating point."""
    L = load()
    cfg = PmrCfg()
    L.pmr_chain_default_cfg(C.byref(cfg))
    cfg.fs_in = fs_in
    cfg.num_channels = num_channels
    cfg.max_block = max_block
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown cfg field %r" % k)
        setattr(cfg, k, int(v) if isinstance(v, bool) else v)
    return cfg


def cfg_design_dict(cfg):
    """Host-only: designed coefficients/integers for cfg (same keys as the oracle's design_dict)."""
    L = load()

    def des(what, idx=0):
        n = L.pmr_cfg_design(C.byref(cfg), what, idx, None, 0)
        out = np.zeros(n, dtype=np.float32)
        L.pmr_cfg_design(C.byref(cfg), what, idx, out.ctypes.data, n)
        return out

    info = lambda what, idx=0: L.pmr_cfg_info(C.byref(cfg), what, idx)
    h = info(INFO_NUM_STAGES)
    return {
        "num_stages": h, "m_stage": [info(INFO_M_STAGE, g) for g in range(h)],
        "hb": [des(DESIGN_HALFBAND, g) for g in range(h)],
        "arb_step": info(INFO_ARB_STEP), "nco_dtheta": info(INFO_NCO_DTHETA), "arb_npfb": info(INFO_ARB_NPFB),
        "arb_m": info(INFO_ARB_M), "arb": des(DESIGN_ARB), "pfb": des(DESIGN_PFB), "pfb_p": info(INFO_PFB_P),
    }


class PmrError(RuntimeError):
    pass


def asgram_ascii(psd_db, nfft, n_transforms, ref=-40.0, div=2.0):
    """asgramcf_execute's character line + peak (host logic of the library): (ascii, peakval_db, peakfreq)."""
    psd = np.ascontiguousarray(psd_db, dtype=np.float32)
    buf = C.create_string_buffer(int(nfft) + 1)
    pv, pf = C.c_float(0), C.c_float(0)
    rc = load().pmr_asgram_ascii(psd.ctypes.data, int(nfft), int(n_transforms), ref, div, buf, C.byref(pv), C.byref(pf))
    if rc:
        raise RuntimeError("pmr_asgram_ascii rc=%d" % rc)
    return buf.raw[:int(nfft)].decode("ascii"), pv.value, pf.value


class _PinnedBlock:
    """Owner of one pmr_host_alloc range: rides on the ctypes buffer the numpy views are made of."""

    def __init__(self, L, p):
        self.L, self.p = L, p

    def __del__(self):
        try:
            self.L.pmr_host_free(self.p)
        except Exception:
            pass


class PmrChain:
    """One IQ stream on one GPU: pmr_chain_create / process_block / reset / destroy."""

    def __init__(self, fs_in=1024000.0, num_channels=16, max_block=100000, audio_gain=4.0, lowpass=False,
                 deemph_fir=False, device=-1, channel_width_hz=12500.0, pfb_m=13, pfb_As=80.0, resamp_As=60.0,
                 dcblock_alpha=0.0005, fm_kf=0.5):
        L = load()
        cfg = PmrCfg()
        L.pmr_chain_default_cfg(C.byref(cfg))
        cfg.fs_in = fs_in
        cfg.num_channels = num_channels
        cfg.channel_width_hz = channel_width_hz
        cfg.max_block = max_block
        cfg.audio_gain = audio_gain
        cfg.lowpass = int(lowpass)
        cfg.deemph_fir = int(deemph_fir)
        cfg.device = device
        cfg.pfb_m = pfb_m
        cfg.pfb_As = pfb_As
        cfg.resamp_As = resamp_As
        cfg.dcblock_alpha = dcblock_alpha
        cfg.fm_kf = fm_kf
        self.cfg = cfg
        self._L = L
        self.h = L.pmr_chain_create(C.byref(cfg))
        if not self.h:
            raise PmrError("pmr_chain_create failed: %s" % (L.pmr_chain_create_error().decode() or "no HIP device, or invalid configuration"))
        self.M = L.pmr_chain_num_channels(self.h)
        self.max_frames = L.pmr_chain_max_frames(self.h)

    # -- lifecycle ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc != PMR_OK:
            raise PmrError("pmr_chain rc=%d: %s" % (rc, self._L.pmr_chain_last_error(self.h).decode()))

    def reset(self):
        self._check(self._L.pmr_chain_reset(self.h))

    def seek(self, n_raw):
        """pmr_chain_seek: zero state at stream position n_raw (as after n_raw zero samples)."""
        self._check(self._L.pmr_chain_seek(self.h, int(n_raw)))

    def position(self):
        """(raw samples consumed, resampled samples produced, frames channelized) since reset / seek."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._L.pmr_chain_position(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def close(self):
        if getattr(self, "h", None):
            self._L.pmr_chain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._L.pmr_chain_synchronize(self.h))

    def set_channel_mask(self, channels=None):
        """Demodulate only `channels` (iterable of indices; None = all): reference semantics, src/sdr_pmr446.c:876-877."""
        if channels is None:
            self._check(self._L.pmr_chain_set_channel_mask(self.h, None, 0))
            return
        words = np.zeros((self.M + 63) // 64, dtype=np.uint64)
        for k in channels:
            words[k >> 6] |= np.uint64(1) << np.uint64(k & 63)
        self._check(self._L.pmr_chain_set_channel_mask(self.h, words.ctypes.data, len(words)))

    def reset_channel(self, k):
        """freqdem_reset + ctcss_detector_reset of channel k (src/sdr_pmr446.c:866-867)."""
        self._check(self._L.pmr_chain_reset_channel(self.h, k))

    def set_overlap(self, on=True):
        self._check(self._L.pmr_chain_set_overlap(self.h, int(on)))

    @property
    def stream(self):
        return self._L.pmr_chain_stream(self.h)

    # -- host-buffer entry point ----------------------------------------------------------------
    def process_block(self, iq, want=("pcm",), fmt=IQ_CF32):
        """iq: complex64 numpy array (fmt IQ_CF32), or an int16 / uint8 array of interleaved I/Q (IQ_CS16 / IQ_CU8: converted on the
        device by the front end -- pmr_chain_process_block_fmt).  Returns dict: n_frames + requested outputs trimmed to n_frames."""
        if fmt == IQ_CF32:
            iq = np.ascontiguousarray(iq, dtype=np.complex64); n_in = len(iq)
        else:
            iq = np.ascontiguousarray(iq, dtype=np.int16 if fmt == IQ_CS16 else np.uint8).reshape(-1); n_in = len(iq) // 2
        want = set(want)
        M, S = self.M, self.max_frames
        pcm = np.zeros((M, S), dtype=np.int16) if "pcm" in want else None
        audio = np.zeros((M, S), dtype=np.float32) if "audio" in want else None
        chan = np.zeros((M, S), dtype=np.complex64) if "chan" in want else None
        rssi = np.zeros(M, dtype=np.float32) if "rssi" in want else None
        if "ctcss" in want:
            self._check(self._L.pmr_chain_ctcss_enable(self.h, 1))
        dbg = bool(want & {"resampled", "fm", "ctcss_lp"})
        if dbg:
            self._check(self._L.pmr_chain_debug_enable(self.h, 1))
        ns = C.c_uint(0)
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.pmr_chain_process_block_fmt(self.h, iq.ctypes.data if n_in else None, fmt, n_in,
                                                        ptr(pcm), ptr(audio), S, C.byref(ns), ptr(chan), ptr(rssi)))
        n = ns.value
        out = {"n_frames": n}
        if pcm is not None:
            out["pcm"] = pcm[:, :n].copy()
        if audio is not None:
            out["audio"] = audio[:, :n].copy()
        if chan is not None:
            out["chan"] = chan[:, :n].copy()
        if rssi is not None:
            out["rssi"] = rssi
        if "ctcss" in want:
            out["ctcss"] = self.ctcss_read()
        if "resampled" in want:
            out["resampled"] = self.debug_read(DEBUG_RESAMPLED, np.complex64)
        if "fm" in want:
            fm = self.debug_read(DEBUG_FM, np.float32)
            out["fm"] = fm.reshape(-1, M).T.copy() if len(fm) else np.zeros((M, 0), np.float32)
        if "ctcss_lp" in want:
            lp = self.debug_read(DEBUG_CTCSS_LP, np.float32)
            out["ctcss_lp"] = lp.reshape(-1, M).T.copy() if len(lp) else np.zeros((M, 0), np.float32)
        return out

    # -- asynchronous host-buffer pair ------------------------------------------------------------
    WANT = {"pcm": 1, "audio": 2, "rssi": 4, "chan": 8}

    def submit_block(self, iq, want=("pcm",), fmt=IQ_CF32):
        """Queue one block.  iq: complex64 array (fmt IQ_CF32), or int16 / uint8 array of interleaved I/Q (IQ_CS16 / IQ_CU8,
        converted on the device); it must stay alive until collected (pinned_array() makes the copy asynchronous)."""
        if fmt == IQ_CF32:
            iq = np.ascontiguousarray(iq, dtype=np.complex64); n = len(iq)
        else:
            iq = np.ascontiguousarray(iq, dtype=np.int16 if fmt == IQ_CS16 else np.uint8).reshape(-1); n = len(iq) // 2
        self._pending = getattr(self, "_pending", [])
        w = sum(self.WANT[k] for k in want)
        self._check(self._L.pmr_chain_submit_block_fmt(self.h, iq.ctypes.data if n else None, fmt, n, w))
        self._pending.append((iq, set(want)))        # only a block the library accepted is waiting to be collected

    def collect_block(self):
        """Outputs of the oldest submitted block: dict like process_block."""
        iq, want = self._pending.pop(0)
        M, S = self.M, self.max_frames
        pcm = np.zeros((M, S), dtype=np.int16) if "pcm" in want else None
        audio = np.zeros((M, S), dtype=np.float32) if "audio" in want else None
        chan = np.zeros((M, S), dtype=np.complex64) if "chan" in want else None
        rssi = np.zeros(M, dtype=np.float32) if "rssi" in want else None
        ns = C.c_uint(0)
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.pmr_chain_collect_block(self.h, ptr(pcm), ptr(audio), S, C.byref(ns), ptr(chan), ptr(rssi)))
        n = ns.value
        out = {"n_frames": n}
        for k, v in (("pcm", pcm), ("audio", audio), ("chan", chan)):
            if v is not None:
                out[k] = v[:, :n].copy()
        if rssi is not None:
            out["rssi"] = rssi
        return out

    def pinned_array(self, n, dtype=np.complex64):
        """numpy view of n elements of memory pinned in the library's HIP runtime (pmr_host_alloc).  The memory lives as long as
        any view of it does (it is NOT freed by close(): a view kept beyond the chain stays valid)."""
        nbytes = int(n) * np.dtype(dtype).itemsize
        p = self._L.pmr_host_alloc(nbytes)
        if not p:
            raise PmrError("pmr_host_alloc failed")
        buf = (C.c_char * nbytes).from_address(p)
        buf._pmr_owner = _PinnedBlock(self._L, p)          # freed when the last view of `buf` is gone
        return np.frombuffer(buf, dtype=dtype)

    # -- device-buffer entry point (bench / zero-copy callers) ----------------------------------
    def process_block_device(self, d_iq, n_in, d_pcm=None, d_audio=None, stride=None, d_chan=None, d_rssi=None):
        """All pointers are integer HIP device addresses (e.g. torch.Tensor.data_ptr()).  Asynchronous."""
        ns = C.c_uint(0)
        self._check(self._L.pmr_chain_process_block_device(self.h, d_iq, n_in, d_pcm, d_audio,
                                                           stride if stride is not None else self.max_frames,
                                                           C.byref(ns), d_chan, d_rssi))
        return ns.value

    def ctcss_enable(self, on=True):
        """Run the CTCSS detector with every following block (pmr_chain_ctcss_enable; `want=("ctcss",)` does this implicitly)."""
        self._check(self._L.pmr_chain_ctcss_enable(self.h, int(bool(on))))

    def ctcss_read(self):
        """CTCSS decisions of the Goertzel blocks completed by the last block: structured array [M][n_events]."""
        cap = self.max_frames // 2441 + 2
        ev = np.zeros((self.M, cap), dtype=CTCSS_EVENT)
        n = C.c_uint(0)
        self._check(self._L.pmr_chain_ctcss_read(self.h, ev.ctypes.data, cap, C.byref(n)))
        return ev[:, :n.value].copy()

    # -- two-step form: squelch decision on THIS block before it is demodulated (reference :828-877) -----
    def channelize_block(self, iq, want=("rssi",)):
        """Front end + channelizer + discriminator + RSSI of one block; the audio part stays pending.  Returns n_frames + rssi / chan."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        M, S = self.M, self.max_frames
        rssi = np.zeros(M, dtype=np.float32) if "rssi" in want else None
        chan = np.zeros((M, S), dtype=np.complex64) if "chan" in want else None
        ns = C.c_uint(0)
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.pmr_chain_channelize_block(self.h, iq.ctypes.data if len(iq) else None, len(iq), C.byref(ns), ptr(chan), S, ptr(rssi)))
        out = {"n_frames": ns.value}
        if rssi is not None:
            out["rssi"] = rssi
        if chan is not None:
            out["chan"] = chan[:, :ns.value].copy()
        return out

    def demodulate_block(self, want=("pcm",)):
        """Audio part of the block channelized last, for the channels the mask has open now."""
        M, S = self.M, self.max_frames
        pcm = np.zeros((M, S), dtype=np.int16) if "pcm" in want else None
        audio = np.zeros((M, S), dtype=np.float32) if "audio" in want else None
        ns = C.c_uint(0)
        ptr = lambda a: a.ctypes.data if a is not None else None
        self._check(self._L.pmr_chain_demodulate_block(self.h, ptr(pcm), ptr(audio), S, C.byref(ns)))
        out = {"n_frames": ns.value}
        if pcm is not None:
            out["pcm"] = pcm[:, :ns.value].copy()
        if audio is not None:
            out["audio"] = audio[:, :ns.value].copy()
        return out

    # -- waterfall line (reference src/sdr_pmr446.c:473-477, :911-915) -----------------
    def spectrum_enable(self, nfft):
        """asgramcf_create(nfft): every following block also yields the averaged periodogram of its resampled samples (0 = off)."""
        self._check(self._L.pmr_chain_spectrum_enable(self.h, int(nfft)))
        self._spec_nfft = int(nfft)

    def spectrum_read(self):
        """(psd_db[4 nfft], n_transforms) of the last block."""
        psd = np.zeros(4 * self._spec_nfft, dtype=np.float32)
        n = C.c_uint(0)
        self._check(self._L.pmr_chain_spectrum_read(self.h, psd.ctypes.data, len(psd), C.byref(n)))
        return psd, n.value

    # -- measurement / introspection -------------------------------------------------------------
    def profile_enable(self, mode=1):
        """0 off, 1 every kernel, m >= 2 only the front-end (roofline) kernel, every (m-1)-th launch (include/pmr_chain.h)."""
        self._check(self._L.pmr_chain_profile_enable(self.h, int(mode)))

    def profile_reset(self):
        self._check(self._L.pmr_chain_profile_reset(self.h))

    def profile(self):
        """{kernel name: (total_ms, launches)} from HIP events on the chain's stream."""
        out = {}
        for i in range(self._L.pmr_chain_profile_count(self.h)):
            ms, n = C.c_double(0), C.c_uint(0)
            self._check(self._L.pmr_chain_profile_get(self.h, i, C.byref(ms), C.byref(n)))
            if n.value:
                out[self._L.pmr_chain_profile_name(self.h, i).decode()] = (ms.value, n.value)
        return out

    def info(self, what, idx=0):
        return self._L.pmr_chain_info(self.h, what, idx)

    def design(self, what, idx=0):
        n = self._L.pmr_chain_design(self.h, what, idx, None, 0)
        out = np.zeros(n, dtype=np.float32)
        self._L.pmr_chain_design(self.h, what, idx, out.ctypes.data, n)
        return out

    def debug_read(self, what, dtype):
        nb = C.c_size_t(0)
        self._check(self._L.pmr_chain_debug_read(self.h, what, None, 0, C.byref(nb)))
        buf = np.zeros(nb.value // np.dtype(dtype).itemsize, dtype=dtype)
        if nb.value:
            self._check(self._L.pmr_chain_debug_read(self.h, what, buf.ctypes.data, nb.value, C.byref(nb)))
        return buf


def make_dsd_cfg(fs_in=1024000.0, sig_rate=12500.0, audio_rate=48000.0, max_block=200000, device=-1):
    cfg = DsdCfg()
    load().pmr_dsd_default_cfg(C.byref(cfg))
    cfg.fs_in, cfg.sig_rate, cfg.audio_rate, cfg.max_block, cfg.device = fs_in, sig_rate, audio_rate, max_block, device
    return cfg


class PmrDsd:
    """The `dsd_in` loop body on the GPU (include/pmr_dsd.h; reference src/dsd_in.c:160-178)."""

    def __init__(self, fs_in=1024000.0, sig_rate=12500.0, audio_rate=48000.0, max_block=200000, device=-1):
        self._L = load()
        self.cfg = make_dsd_cfg(fs_in, sig_rate, audio_rate, max_block, device)
        self.h = self._L.pmr_dsd_create(C.byref(self.cfg))
        if not self.h:
            raise PmrError("pmr_dsd_create failed (no HIP device, or invalid configuration)")
        self.max_out = self._L.pmr_dsd_max_out(self.h)

    def _check(self, rc):
        if rc != 0:
            raise PmrError("pmr_dsd rc=%d: %s" % (rc, self._L.pmr_dsd_last_error(self.h).decode()))

    def reset(self):
        self._check(self._L.pmr_dsd_reset(self.h))

    def synchronize(self):
        self._check(self._L.pmr_dsd_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None):
            self._L.pmr_dsd_destroy(self.h)
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
        audio = np.zeros(cap, dtype=np.float32) if "audio" in want else None
        nz = C.c_uint(0)
        self._check(self._L.pmr_dsd_process_block(self.h, iq.ctypes.data, len(iq), pcm.ctypes.data,
                                                  audio.ctypes.data if audio is not None else None, cap, C.byref(nz)))
        out = {"n_out": nz.value, "pcm": pcm[:nz.value].copy()}
        if audio is not None:
            out["audio"] = audio[:nz.value].copy()
        if "resampled" in want:
            out["resampled"] = self.debug_read(0, np.complex64)
        if "fm" in want:
            out["fm"] = self.debug_read(1, np.float32)
        return out

    def process_block_device(self, d_iq, n_in, d_pcm=None, d_audio=None, cap=0):
        nz = C.c_uint(0)
        self._check(self._L.pmr_dsd_process_block_device(self.h, d_iq, n_in, d_pcm, d_audio, cap, C.byref(nz)))
        return nz.value

    def debug_read(self, what, dtype):
        nb = C.c_size_t(0)
        self._check(self._L.pmr_dsd_debug_read(self.h, what, None, 0, C.byref(nb)))
        buf = np.zeros(nb.value // np.dtype(dtype).itemsize, dtype=dtype)
        if nb.value:
            self._check(self._L.pmr_dsd_debug_read(self.h, what, buf.ctypes.data, nb.value, C.byref(nb)))
        return buf


WAV_F32, WAV_S16, RAW_S16 = 0, 1, 2


class IqReader:
    """pmr_iq_reader (include/pmr_io.h): recorded IQ as cf32 blocks, the stand-in for readStream."""

    def __init__(self, path, fmt=IQ_CF32):
        self._L = load()
        self.h = self._L.pmr_iq_reader_open(os.fsencode(path), fmt)
        if not self.h:
            raise PmrError("pmr_iq_reader_open failed: %s" % path)

    def read(self, max_samples):
        buf = np.zeros(max_samples, dtype=np.complex64)
        n = self._L.pmr_iq_reader_read(self.h, buf.ctypes.data, max_samples)
        if n < 0:
            raise PmrError("pmr_iq_reader_read rc=%d" % n)
        return buf[:n]

    def close(self):
        if self.h:
            self._L.pmr_iq_reader_close(self.h)
            self.h = None


class WavWriter:
    """pmr_wav_writer (include/pmr_io.h): planar [channels][stride] in, interleaved WAV / raw s16 out."""

    def __init__(self, path, fmt, sample_rate, channels=1):
        self._L = load()
        self.fmt, self.channels = fmt, channels
        self.h = self._L.pmr_wav_writer_open(os.fsencode(path), fmt, sample_rate, channels)
        if not self.h:
            raise PmrError("pmr_wav_writer_open failed: %s" % path)

    def write(self, data):
        data = np.atleast_2d(data)
        assert data.shape[0] == self.channels
        if self.fmt == WAV_F32:
            d = np.ascontiguousarray(data, dtype=np.float32)
            rc = self._L.pmr_wav_writer_write_f32(self.h, d.ctypes.data, d.shape[1], d.shape[1])
        else:
            d = np.ascontiguousarray(data, dtype=np.int16)
            rc = self._L.pmr_wav_writer_write_s16(self.h, d.ctypes.data, d.shape[1], d.shape[1])
        if rc != 0:
            raise PmrError("pmr_wav_writer_write rc=%d" % rc)

    def close(self):
        if self.h:
            rc = self._L.pmr_wav_writer_close(self.h)
            self.h = None
            if rc != 0:
                raise PmrError("pmr_wav_writer_close rc=%d" % rc)
