"""bench.py prints a headline only from the in-tree PRODUCT build: a library selected by PMR_LIBRARY is refused
before anything touches a GPU, `--allow-experiment` is what the A/B tools pass, and the library itself says whether it was compiled with
the experiment gate open (pmr_chain_info(NULL, PMR_INFO_EXPERIMENT_BUILD); csrc/pmr_experiment.h)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_library_selected_by_the_environment():
    from sdr_pmr446_amd import build
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PMR_LIBRARY"] = build.LIB                       # even the product's own file: the headline never honours the variable
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to print a headline" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_product_build_reports_no_experiment_and_defining_a_hook_without_the_gate_does_not_compile(tmp_path):
    from sdr_pmr446_amd import build, chain
    L = chain.load()
    assert L.pmr_chain_info(None, 11, 0) == 0            # PMR_INFO_EXPERIMENT_BUILD, the only query that needs no handle
    src = os.path.join(build.CSRC, "pmr_squelch.c")      # any unit that includes pmr_kernels.h would do; gcc is enough for this one
    probe = tmp_path / "probe.c"
    probe.write_text('#include "%s"\nint main(void) { return 0; }\n' % os.path.join(build.CSRC, "pmr_experiment.h"))
    ok = subprocess.run(["gcc", "-fsyntax-only", str(probe)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    for hook in ("-DEXP_SKIP_FIR", "-DFE_STOP=1", "-DFE_S1_LDS", "-DEXP_L2_INLINE=128", "-DCW_NT=128"):
        bad = subprocess.run(["gcc", "-fsyntax-only", hook, str(probe)], capture_output=True, text=True)
        assert bad.returncode != 0 and "PMR_EXPERIMENT" in bad.stderr, hook
        gated = subprocess.run(["gcc", "-fsyntax-only", hook, "-DPMR_EXPERIMENT", str(probe)], capture_output=True, text=True)
        assert gated.returncode == 0, (hook, gated.stderr)
    assert os.path.exists(src)


def test_the_product_build_takes_no_flags_from_the_environment():
    from sdr_pmr446_amd import build
    env = dict(os.environ, PMR_HIPCC_FLAGS="-DEXP_SKIP_FIR")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sdr_pmr446_amd", "build.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--variant" in r.stderr
