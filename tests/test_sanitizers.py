"""CPU-tier sanitizers (the reference builds with none, CMakeLists.txt:15-16): a slice of tools/asan_tier.sh inside the test
suite -- the oracle and the product's host C units compiled with -fsanitize=address,undefined, the host-logic tests run on them with the
sanitizer runtimes preloaded.  `bash tools/asan_tier.sh` runs the whole tier (76 tests)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.exists(subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()),
                    reason="gcc's libasan is not installed")
def test_host_c_and_oracle_are_clean_under_asan_and_ubsan():
    env = {k: v for k, v in os.environ.items() if not k.startswith(("PMR_", "LD_PRELOAD", "ASAN_", "UBSAN_"))}
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_tier.sh"), "--quick"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "sanitized tier: libasan mapped=1 oracle=1 product=1" in r.stdout
    assert " passed" in r.stdout and "failed" not in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
