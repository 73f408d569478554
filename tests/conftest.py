import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "nopoison: GPU test that measures time (runs without the LDS / scratch poison mode)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not _have_gpu():
        pytest.skip("no GPU in this container")


# ---- order of the -m gpu tier --------------------------------------------------------------------------------------------------
# The driver runs `pytest -x`: everything behind the first failure is un-tested.  So the hot path (SURVEY s8 rows a0-a10 on every
# BASELINE config: goldens, full-bank +-1 LSB vs the oracle through the un-synchronised device entry on cfg2 / cfg3 / cfg5, the
# path bench.py times) is judged FIRST; then the "next" rows in SURVEY's order f1 (mask / squelch), f2 (CTCSS), f3 (dsd_in),
# f4 (ingest / egress / waterfall); fallback kernels and the soak last.  First matching pattern wins; ties keep file order.
_ORDER = [
    ("test_golden.py::test_hip_matches_golden", 0),
    ("test_golden.py::test_hip_rssi", 20),                      # f1
    ("test_gpu_fullbank.py::test_every_channel_loaded", 1),
    ("test_gpu_pipelined.py::test_bench_path_unsynchronised_full_size_blocks", 2),
    ("test_gpu_carry.py::test_carry_at_load_equals_in_place_and_oracle", 3),
    ("test_gpu_parity.py::test_cfg5_1024_channels", 4),
    ("test_gpu_parity.py::test_cfg3_256_channels", 4),
    ("test_gpu_parity.py::test_cfg2_one_reference_block", 4),
    ("test_gpu_parity.py::test_reference_operating_point", 4),
    ("test_gpu_parity.py", 10),
    ("test_gpu_fullbank.py", 11),
    ("test_gpu_carry.py", 12),
    ("test_gpu_pipelined.py", 13),
    ("test_gpu_synth.py", 14),
    ("test_gpu_streams.py", 14),                                # the streams the other ranks of bench.py --gpus N run (cfg4)
    ("test_seek.py", 16),                                       # counters beyond 2^32
    ("test_gpu_device.py", 16),
    ("test_gpu_hostpath.py::test_integer_ingest", 50),           # f4
    ("test_gpu_hostpath.py::test_sync_integer", 50),
    ("test_gpu_hostpath.py::test_two_step", 20),                # f1: the call split where the squelch sits
    ("test_gpu_hostpath.py::test_rssi", 20),
    ("test_gpu_hostpath.py", 15),
    ("test_gpu_mask.py", 21),                                   # f1
    ("test_gpu_ctcss.py", 30),                                  # f2
    ("test_gpu_dsd.py", 40),                                    # f3
    ("test_io.py", 51),                                         # f4
    ("test_spectrum.py", 52),
    ("test_gpu_bench_ranks.py", 55),                            # bench.py --gpus 2: the cfg4 path with every rank's parity
    ("test_gpu_poison.py", 60),
    ("test_gpu_build_identity.py", 75),                         # builds the round-5 kernel form on the box: bit-identical outputs
    ("test_gpu_variants.py", 80),
    ("test_gpu_soak.py", 90),
]


def _rank(nodeid):
    for pat, r in _ORDER:
        if pat in nodeid:
            return r
    return 70


def pytest_collection_modifyitems(config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu:
        return
    rest = [it for it in items if not it.get_closest_marker("gpu")]
    gpu.sort(key=lambda it: _rank(it.nodeid))                   # stable: file order inside a rank
    items[:] = rest + gpu


# ---- poison mode for the -m gpu tier (include/pmr_chain.h pmr_debug_poison, csrc/pmr_poison.hip) ---------------------------------
# Every GPU test runs with ALL LDS of every CU overwritten by signalling NaNs before each kernel launch of the library and with
# the library's scratch buffers (and pmr_device_alloc's) filled with 0xFF instead of zeros: results that depend on bytes nobody
# wrote fail on every box (round 3's red driver run was such a read, green on three boxes).  PMR_TEST_POISON=0 switches it off;
# tests marked `nopoison` (they measure time) run without it.
@pytest.fixture(autouse=True)
def _gpu_poison(request):
    if request.node.get_closest_marker("gpu") is None or not _have_gpu():
        yield
        return
    from sdr_pmr446_amd import chain
    L = chain.load()
    on = os.environ.get("PMR_TEST_POISON", "1") != "0" and request.node.get_closest_marker("nopoison") is None
    was = L.pmr_debug_poison(1 if on else 0)
    env_was = os.environ.get("PMR_DEBUG_POISON")
    os.environ["PMR_DEBUG_POISON"] = "1" if on else "0"        # child processes (examples/pmr446_file) inherit the mode
    yield
    L.pmr_debug_poison(was)
    if env_was is None:
        os.environ.pop("PMR_DEBUG_POISON", None)
    else:
        os.environ["PMR_DEBUG_POISON"] = env_was
