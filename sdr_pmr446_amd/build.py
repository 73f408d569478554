"""In-tree build of libpmr446_hip.so (gfx950) -- host C with gcc, kernels with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpmr446_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

C_SOURCES = ["pmr_chain.c", "pmr_design.c", "pmr_squelch.c", "pmr_dsd.c", "pmr_io.c"]
HIP_SOURCES = ["pmr_kernels.hip", "pmr_frontend.hip", "pmr_fe_fast.hip", "pmr_channelize_small.hip", "pmr_channelize_wide.hip", "pmr_fir_mfma4.hip", "pmr_fir_fft.hip", "pmr_ctcss.hip", "pmr_synth.hip", "pmr_spectrum.hip",
               "pmr_dsd_kernels.hip", "pmr_poison.hip"]
EXTRA_HIP_FLAGS = os.environ.get("PMR_HIPCC_FLAGS", "-fno-slp-vectorize").split()
EXTRA_C_FLAGS = os.environ.get("PMR_CC_FLAGS", "").split()          # experiment builds only (tools/variant_bench.sh)
HEADERS = ["pmr_design.h", "pmr_kernels.h", "pmr_internal.h", "pmr_fe_common.hpp", "pmr_carry_load.hpp", os.path.join("..", "..", "include", "pmr_chain.h"),
           os.path.join("..", "..", "include", "pmr_dsd.h"), os.path.join("..", "..", "include", "pmr_io.h"), os.path.join("..", "..", "include", "pmr_mem.h"),
           os.path.join("..", "data", "pmr446_taps.h")]


def kernel_sources_sha256():
    """sha256 over the kernel sources (csrc/*.hip, *.hpp and the kernels' shared header), names and contents in sorted order: what
    ties a committed PMC measurement (profiles/traffic.json) to the kernels it was taken with -- bench.py marks roofline.traffic
    stale when the tree's hash differs from the entry's."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")) or f == "pmr_kernels.h":
            h.update(f.encode() + b"\0")
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
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


def build_variant(name, hip_flags="", c_flags=None, verbose=False):
    """Another BUILD of the library for same-box A/B runs (tools/ab_libs.py, PMR_LIBRARY): build_ab/<name>/libpmr446_hip.so, compiled
    with extra -D flags (they reach hipcc AND gcc unless c_flags is given), objects beside it.  Not part of the product."""
    out = os.path.join(os.path.dirname(HERE), "build_ab", name)
    os.makedirs(out, exist_ok=True)
    if c_flags is None:
        c_flags = " ".join(t for t in hip_flags.split() if t.startswith("-D"))
    lib = os.path.join(out, "libpmr446_hip.so")
    _compile(lib, out, ["-fno-slp-vectorize"] + hip_flags.split(), c_flags.split(), verbose)
    return lib


def build(force=False, verbose=False):
    """Compile for gfx950 (cross-compiles without a GPU).  Returns the library path."""
    if not force and not _stale():
        return LIB
    _compile(LIB, CSRC, EXTRA_HIP_FLAGS, EXTRA_C_FLAGS, verbose)
    build_example(verbose)
    return LIB


def _compile(lib, objdir, hip_flags, c_flags, verbose):
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objs = []
    for f in C_SOURCES:
        o = os.path.join(objdir, f[:-2] + ".o")
        cmd = ["gcc", "-std=gnu11", "-O2", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter",
               "-I" + os.path.join(ROCM, "include")] + c_flags + ["-c", os.path.join(CSRC, f), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    for f in HIP_SOURCES:
        o = os.path.join(objdir, f[:-4] + ".o")
        # -fno-slp-vectorize: keep f32 FMAs as v_fma/v_fmac; hipcc otherwise SLP-packs adjacent ones into
        # v_pk_fma_f32, which is slower than two plain FMAs on gfx950 (measured on k_fir_*; MI355X_MICROARCH.md)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-parameter"] + \
              hip_flags + ["-c", os.path.join(CSRC, f), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lm", "-lpthread"]
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
    elif "--variant" in sys.argv:                       # build.py --variant NAME "-DFOO -DBAR=1"
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2] if len(sys.argv) > i + 2 else ""))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
