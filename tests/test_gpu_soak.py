"""Randomised GPU-vs-oracle soak (tools/soak.py): random (fs_in, M) incl. M = 4...512, filter options, ragged / tiny / empty
block splits -- every case must keep int16 PCM within +-1 LSB of the oracle on the channels that carry a signal."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomised_soak_20s():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "20", "3"], capture_output=True, text=True,
                       timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0, tail
    assert tail.startswith("soak:") and "worst |pcm diff| = " in tail, tail
    assert int(tail.split("soak:")[1].split("cases")[0]) >= 20, tail


@pytest.mark.gpu
def test_pipelined_soak_20s():
    """tools/soak_pipelined.py: random configurations (cfg2 / cfg3 / cfg5 / reference point / 64 channels), random ragged block
    sequences incl. empty and tiny blocks, CTCSS detector / channel mask / waterfall randomly on: un-synchronised pipelined
    calls vs the same calls synchronised one by one -- PCM, CTCSS events and PSD bit-identical."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_pipelined.py"), "20", "11"], capture_output=True,
                       text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0, tail
    assert tail.startswith("soak_pipelined:") and "bit-identical" in tail, tail
    assert int(tail.split("soak_pipelined:")[1].split("cases")[0]) >= 50, tail
