#!/bin/bash
# CPU-tier sanitizers (SURVEY s5 "ASan/UBSan on the C restatement").  Runs HERE, without a GPU:
#   1. the oracle under AddressSanitizer + UBSan (make -C oracle asan) through every CPU test that drives it;
#   2. the product's gcc-compiled host C units (pmr_chain.c planning / ring bookkeeping, pmr_design.c, pmr_squelch.c, pmr_io.c, pmr_dsd.c)
#      built the same way (build.py --asan -> build_ab/asan/libpmr446_hip.so; the kernels are compiled as always and never run here)
#      through the host-logic tests: design, plan, squelch, I/O, seek arithmetic, C-ABI symbols.
# python itself is not instrumented, so the sanitizer runtimes are preloaded; leak checking is off (the interpreter leaks by design).
# Exit code: pytest's.  Usage: bash tools/asan_tier.sh [extra pytest arguments]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
make -C oracle -s asan
python3 sdr_pmr446_amd/build.py --asan > /dev/null
ASAN_SO=$(gcc -print-file-name=libasan.so); UBSAN_SO=$(gcc -print-file-name=libubsan.so)
export LD_PRELOAD="$ASAN_SO $UBSAN_SO"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export PMR_ORACLE_LIB="$R/oracle/liboracle_pmr_asan.so" PMR_LIBRARY="$R/build_ab/asan/libpmr446_hip.so" PMR_NO_TORCH=1
# proof that the instrumented libraries are the ones the tests get (and that the runtime is mapped): one line on stdout
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import oracle.binding as ob
from sdr_pmr446_amd import chain
ob.lib(); chain.load()
maps = open("/proc/self/maps").read()
print("sanitized tier: libasan mapped=%d oracle=%d product=%d" % ("libasan" in maps, "liboracle_pmr_asan.so" in maps, "build_ab/asan/libpmr446_hip.so" in maps))
assert "libasan" in maps and "liboracle_pmr_asan.so" in maps and "build_ab/asan/libpmr446_hip.so" in maps
PY
QUICK=0; if [ "$1" = "--quick" ]; then QUICK=1; shift; fi
if [ $QUICK = 1 ]; then      # tests/test_sanitizers.py: a slice of the tier inside the CPU test suite (~15 s)
  exec python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider tests/test_oracle_pins.py tests/test_host_logic.py tests/test_squelch.py tests/test_seek.py "$@"
fi
# (torch is never imported here: its bundled runtimes and ASan's interposed allocator do not get along, and no test below needs it)
exec python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider \
    tests/test_golden.py tests/test_oracle_pins.py tests/test_oracle_model.py tests/test_ref_fixtures.py tests/test_dsd_cpu.py \
    tests/test_parity_rule.py tests/test_host_logic.py tests/test_squelch.py tests/test_io.py tests/test_seek.py tests/test_spectrum.py "$@"
