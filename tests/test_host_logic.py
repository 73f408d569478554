"""CPU tier for the product: the C-ABI library loads and exports every symbol include/pmr_chain.h declares, its
host-side design equals the oracle's bit for bit, its closed-form block planner reproduces the oracle's per-block
counts, and it fails loudly (no CPU fallback) when no HIP device exists.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from parity_util import CFG2, CFG3, CFG5, CFG_REF
from sdr_pmr446_amd import chain

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = chain.load()
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("pmr_chain.h", "pmr_dsd.h", "pmr_io.h", "pmr_mem.h"))
    declared = set(re.findall(r"\b(pmr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(chain.ABI_SYMBOLS), declared ^ set(chain.ABI_SYMBOLS)
    for sym in declared:
        assert getattr(L, sym) is not None


@pytest.mark.parametrize("fs,M", [CFG_REF, CFG2, CFG3, CFG5])
def test_host_design_is_bit_identical_to_oracle(fs, M):
    d = chain.cfg_design_dict(chain.make_cfg(fs, M, 100000))
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=100000).design_dict()
    for key in ("num_stages", "m_stage", "arb_step", "nco_dtheta", "arb_npfb", "arb_m", "pfb_p"):
        assert d[key] == o[key], key
    for a, b in zip(d["hb"], o["hb"]):
        assert np.array_equal(a, b)
    assert np.array_equal(d["arb"], o["arb"]) and np.array_equal(d["pfb"], o["pfb"])


def test_max_frames_rule():
    L = chain.load()
    assert L.pmr_cfg_max_frames(C.byref(chain.make_cfg(1024000.0, 16, 100000))) == 2441   # src/sdr_pmr446.c:37,736


@pytest.mark.parametrize("fs,M", [CFG_REF, CFG2, CFG3])
def test_block_planner_matches_oracle_counts(fs, M):
    """The launch-sizing arithmetic (ny, ns per block) against the oracle's actual per-block outputs."""
    L = chain.load()
    cfg = chain.make_cfg(fs, M, 70000)
    st = chain.PlanState(0, 0, 0)
    o = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=70000)
    rng = np.random.default_rng(3)
    for i in range(60):
        n = int(rng.integers(0, 70000)) if i % 7 else int(rng.integers(0, 40))
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.1
        r = o.process_block(x, want=("resampled",))
        ny, ns = C.c_uint(0), C.c_uint(0)
        assert L.pmr_cfg_plan_block(C.byref(cfg), C.byref(st), n, C.byref(ny), C.byref(ns)) == 0
        assert (ny.value, ns.value) == (len(r["resampled"]), r["n_frames"]), (i, n)


def test_invalid_configurations_are_rejected():
    L = chain.load()
    for kw in (dict(num_channels=12), dict(num_channels=0), dict(fs_in=1000.0), dict(pfb_m=0)):
        cfg = chain.make_cfg(**{**dict(fs_in=1024000.0, num_channels=16, max_block=1000), **kw})
        assert L.pmr_cfg_info(C.byref(cfg), chain.INFO_NUM_STAGES, 0) == 0
        assert L.pmr_cfg_max_frames(C.byref(cfg)) == 0


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(chain.PmrError):
        chain.PmrChain()
