"""In-tree build of libpmr446_hip.so (gfx950) -- host C with gcc, kernels with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpmr446_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

C_SOURCES = ["pmr_chain.c", "pmr_chain_plan.c", "pmr_chain_frontend.c", "pmr_chain_host.c", "pmr_chain_aux.c", "pmr_design.c", "pmr_squelch.c", "pmr_dsd.c", "pmr_io.c"]
HIP_SOURCES = ["pmr_kernels.hip", "pmr_frontend.hip", "pmr_fe_fast.hip", "pmr_channelize_small.hip", "pmr_channelize_wide.hip", "pmr_fir_mfma4.hip", "pmr_fir_fft.hip", "pmr_ctcss.hip", "pmr_synth.hip", "pmr_spectrum.hip",
               "pmr_dsd_kernels.hip", "pmr_poison.hip"]
# -fno-slp-vectorize: see _compile.  The PRODUCT build takes no other flags: anything else is a variant (build_variant ->
# build_ab/NAME/, -DPMR_EXPERIMENT added unless it is a sanitizer build of unchanged sources), so the in-tree library can never be
# an experiment build by accident (tools/variant_*.sh used to rebuild it in place with --force)
PRODUCT_HIP_FLAGS = ["-fno-slp-vectorize"]
PRODUCT_C_FLAGS = []
SANITIZE_C_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g", "-O1"]
HEADERS = ["pmr_design.h", "pmr_kernels.h", "pmr_experiment.h", "pmr_internal.h", "pmr_chain_priv.h", "pmr_fe_common.hpp", "pmr_carry_load.hpp", os.path.join("..", "..", "include", "pmr_chain.h"),
           os.path.join("..", "..", "include", "pmr_dsd.h"), os.path.join("..", "..", "include", "pmr_io.h"), os.path.join("..", "..", "include", "pmr_mem.h"),
           os.path.join("..", "data", "pmr446_taps.h")]


def kernel_sources_sha256():
    """sha256 over everything that decides what a launch moves through HBM: the kernel sources (csrc/*.hip, *.hpp, the shared headers),
    the HOST PLAN (pmr_chain.c: which kernels run, tile paddings, transform sizes, streams; pmr_design.c; the tap tables) and the
    product's compiler flags -- names and contents in sorted order.  It ties a committed PMC measurement (profiles/traffic.json) to
    the tree it was taken with: bench.py marks roofline.traffic stale when the tree's hash differs from the entry's.  (Round 5 hashed
    the kernels only; a plan change left a stale number marked fresh -- ADVICE r05.)"""
    import hashlib
    h = hashlib.sha256()
    # every kernel unit and shared header, and the host units that make the main chain's plan (pmr_chain*.c, pmr_design.c); NOT the
    # host-only consumers that cannot change a launch of it (squelch logic, file I/O, the dsd_in host side)
    names = [f for f in sorted(os.listdir(CSRC))
             if f.endswith((".hip", ".hpp", ".h")) or (f.endswith(".c") and (f.startswith("pmr_chain") or f == "pmr_design.c"))]
    for f in names:
        h.update(f.encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(HERE, "data", "pmr446_taps.h"), "rb") as fh:
        h.update(b"pmr446_taps.h\0" + fh.read())
    h.update(" ".join(PRODUCT_HIP_FLAGS + ["|"] + PRODUCT_C_FLAGS).encode())
    return h.hexdigest()


def library_sha256(path=None):
    """sha256 of the built library file (bench.py's `library` record: which binary produced the line)."""
    import hashlib
    h = hashlib.sha256()
    with open(path or LIB, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in C_SOURCES + HIP_SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_variant(name, hip_flags="", c_flags=None, verbose=False, experiment=True):
    """Another BUILD of the library for same-box A/B runs (tools/ab_libs.py, PMR_LIBRARY): build_ab/<name>/libpmr446_hip.so, compiled
    with extra -D flags (they reach hipcc AND gcc unless c_flags is given), objects beside it.  Not part of the product: every such
    build gets -DPMR_EXPERIMENT (csrc/pmr_experiment.h: the one gate of the hooks; the library then reports
    PMR_INFO_EXPERIMENT_BUILD = 1 and bench.py refuses to print a headline from it)."""
    out = os.path.join(os.path.dirname(HERE), "build_ab", name)
    os.makedirs(out, exist_ok=True)
    if c_flags is None:
        c_flags = " ".join(t for t in hip_flags.split() if t.startswith("-D"))
    gate = ["-DPMR_EXPERIMENT"] if experiment else []
    lib = os.path.join(out, "libpmr446_hip.so")
    _compile(lib, out, PRODUCT_HIP_FLAGS + gate + hip_flags.split(), PRODUCT_C_FLAGS + gate + c_flags.split(), verbose)
    return lib


def build_sanitized(verbose=False):
    """The product's sources with the HOST C units (planning, rings' bookkeeping, squelch, I/O, dsd host side) under
    AddressSanitizer + UBSan: build_ab/asan/libpmr446_hip.so.  CPU tier only (tools/asan_tier.sh runs the host-logic tests on it with
    libasan preloaded); the kernels are compiled as always -- GPU sanitizers are not available on the pool.  Not an experiment
    build: no hook is defined, results are the product's."""
    return build_variant("asan", "", " ".join(SANITIZE_C_FLAGS), verbose, experiment=False)


def build(force=False, verbose=False):
    """Compile the PRODUCT for gfx950 (cross-compiles without a GPU).  Returns the library path."""
    for v in ("PMR_HIPCC_FLAGS", "PMR_CC_FLAGS"):
        if os.environ.get(v):
            raise RuntimeError("%s is set: the in-tree library is the product and takes no extra flags; build a variant instead "
                               "(python3 sdr_pmr446_amd/build.py --variant NAME \"%s\" -> build_ab/NAME/)" % (v, os.environ[v]))
    if not force and not _stale():
        return LIB
    _compile(LIB, CSRC, PRODUCT_HIP_FLAGS, PRODUCT_C_FLAGS, verbose)
    build_example(verbose)
    return LIB


def _compile(lib, objdir, hip_flags, c_flags, verbose):
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objs, cmds = [], []
    for f in C_SOURCES:
        o = os.path.join(objdir, f[:-2] + ".o")
        cmds.append(["gcc", "-std=gnu11", "-O2", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter",
                     "-I" + os.path.join(ROCM, "include")] + c_flags + ["-c", os.path.join(CSRC, f), "-o", o])
        objs.append(o)
    for f in HIP_SOURCES:
        o = os.path.join(objdir, f[:-4] + ".o")
        # -fno-slp-vectorize: keep f32 FMAs as v_fma/v_fmac; hipcc otherwise SLP-packs adjacent ones into
        # v_pk_fma_f32, which is slower than two plain FMAs on gfx950 (measured on k_fir_*; MI355X_MICROARCH.md)
        cmds.append([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-parameter"] +
                    hip_flags + ["-c", os.path.join(CSRC, f), "-o", o])
        objs.append(o)

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the units are independent: compile them side by side (a variant build on the GPU box costs box time)
    with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as ex:
        list(ex.map(run, cmds))
    sanitized = any(t.startswith("-fsanitize") for t in c_flags)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lm", "-lpthread"]
    if sanitized:            # gcc's sanitizer runtimes are preloaded by whoever loads the library (tools/asan_tier.sh)
        cmd += ["-Wl,--allow-shlib-undefined"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


EXAMPLE = os.path.join(HERE, "pmr446_file")
EXAMPLE_THREADS = os.path.join(HERE, "pmr446_threads")


def _build_c_example(src, exe, extra, verbose):
    root = os.path.dirname(HERE)
    cmd = ["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", src), "-o", exe, "-L" + HERE, "-lpmr446_hip",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib")] + extra
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return exe


def build_example(verbose=False):
    """examples/pmr446_file.c: headless file -> WAV / s16 harness, linked against the in-tree library;
    examples/pmr446_threads.c: one pthread per handle (the deployment model of include/pmr_chain.h), self-checking."""
    _build_c_example("pmr446_threads.c", EXAMPLE_THREADS, ["-lpthread"], verbose)
    return _build_c_example("pmr446_file.c", EXAMPLE, [], verbose)


if __name__ == "__main__":
    if "--kernel-hash" in sys.argv:
        print(kernel_sources_sha256())
    elif "--asan" in sys.argv:                          # host C units under ASan + UBSan -> build_ab/asan/ (tools/asan_tier.sh)
        print(build_sanitized(verbose="--verbose" in sys.argv))
    elif "--variant" in sys.argv:                       # build.py --variant NAME "-DFOO -DBAR=1"
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2] if len(sys.argv) > i + 2 else ""))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
