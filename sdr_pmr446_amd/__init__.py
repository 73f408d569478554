"""sdr_pmr446_amd -- MI355X-native (gfx950) per-block IQ DSP chain of mryndzionek/sdr_pmr446.

Only the hot path lives here: csrc/ holds the HIP kernels and the C-ABI
(include/pmr_chain.h); chain.py is the host-side mirror that binds the C-ABI with ctypes;
synth.py is the synthetic cf32 source that stands in for the SoapySDR ingest.
"""
__all__ = ["chain", "synth", "build"]
