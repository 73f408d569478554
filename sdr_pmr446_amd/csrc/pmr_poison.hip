// pmr_poison.hip -- TEST-ONLY poison mode of libpmr446_hip.so (pmr_debug_poison, include/pmr_chain.h).
//
// Why it exists: round 3's CTCSS detector read LDS beyond its workgroup's allocation for EMPTY segments and multiplied what it
// found by zero.  Whatever the previous workgroup on that CU had left there decided the result: zeros and ordinary floats passed,
// two int16 PCM samples that happen to be the bit pattern of a NaN did not -- green on three boxes, red on the fourth.  Zero-filled
// device buffers and never-poisoned LDS make every "reads garbage, multiplies by zero" and every read-before-write invisible.
//
// With the mode on:
//  * BEFORE every kernel launch of this library (PMR_KLAUNCH, pmr_kernels.h) a poison kernel runs on the same stream: one
//    workgroup per CU, each owning the CU's whole LDS, writes a signalling-NaN pattern (0x7FA0DEAD: NaN as f32, NaN as two
//    bf16 / f16 halves' high part, a huge value as int16 pairs) over all of it.  The kernels of one stream are ordered, so the
//    kernel under test starts on CUs whose LDS holds nothing but NaNs;
//  * every scratch buffer pmr_chain.c allocates is filled with 0xFF bytes (NaN as f32, -1 as int) instead of zeros; buffers that
//    hold carried STATE are zeroed as always (dev_alloc_state).
// A kernel whose result depends on bytes it did not write then fails deterministically, on every box.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <atomic>

#include "pmr_kernels.h"

#define PMR_POISON_WORD 0x7FA0DEADu

// process-wide, read by every launch of every handle's thread: an atomic (two threads that find it unread both read the environment
// and store the same value)
static std::atomic<int> g_poison{-1};                              // -1: not read yet (environment PMR_DEBUG_POISON)

extern "C" int pmr_debug_poison_enabled(void)
{
    int v = g_poison.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("PMR_DEBUG_POISON");
        v = (e && e[0] && e[0] != '0') ? 1 : 0;
        int expect = -1;
        if (!g_poison.compare_exchange_strong(expect, v, std::memory_order_relaxed)) v = expect;   // pmr_debug_poison() got there first
    }
    return v;
}

// the kernel units' half of pmr_chain_info(.., PMR_INFO_EXPERIMENT_BUILD, ..): every .hip unit gets the same flags (build.py)
extern "C" int pmr_kernels_experiment_build(void) { return PMR_EXPERIMENT_BUILD; }

extern "C" int pmr_debug_poison(int on)
{
    const int was = pmr_debug_poison_enabled();
    g_poison.store(on ? 1 : 0, std::memory_order_relaxed);
    return was;
}

// every thread writes its share of the workgroup's LDS, then the workgroup idles long enough (~20 us) for the dispatcher to have
// placed one workgroup on EVERY CU (each takes the whole LDS: two never share a CU, none can finish early and free a CU for a
// second one while others are still waiting to be placed)
__global__ __launch_bounds__(1024) void k_poison_lds(unsigned words, unsigned word)
{
    extern __shared__ unsigned pz[];
    for (unsigned i = threadIdx.x; i < words; i += 1024u) pz[i] = word;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 40000ull) __builtin_amdgcn_s_sleep(32);
    // keep the stores alive: read one word back
    if (pz[(threadIdx.x * 61u) % words] != word) __builtin_trap();
}

// The whole 160 KiB of a gfx950 CU's LDS is requested EXPLICITLY, whatever hipDeviceAttributeMaxSharedMemoryPerBlock reports (other
// gfx9 parts say 64 KiB: two 64 KiB workgroups per CU would leave 32 KiB of every CU unpoisoned and the one-per-CU placement
// argument above would not hold).  A runtime that does not grant it makes the launch FAIL (and with it every test of the tier):
// the mode never degrades silently.  The per-device "limit raised" flags are an atomic word (pmr_attr_flags), like every other
// launcher's: handles on different devices may be driven from different threads.
#define PMR_POISON_LDS_BYTES (160 * 1024)
extern "C" int pmr_debug_poison_lds(pmr_stream_t s)
{
    static pmr_attr_flags raised;
    static std::atomic<int> dev_cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (pmr_attr_needed(raised)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 PMR_POISON_LDS_BYTES);
        if (e != hipSuccess) { raised.fetch_and(~(1ull << dev), std::memory_order_relaxed); return (int)e; }
    }
    int cus = dev_cus[dev].load(std::memory_order_relaxed);
    if (!cus) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
        dev_cus[dev].store(cus, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(k_poison_lds, dim3((unsigned)cus), dim3(1024), (size_t)PMR_POISON_LDS_BYTES, (hipStream_t)s,
                       (unsigned)PMR_POISON_LDS_BYTES / 4u, PMR_POISON_WORD);
    return (int)hipGetLastError();
}

// what does a freshly started workgroup find in its LDS?  Copies the first `words` words of each workgroup's (uninitialised)
// dynamic LDS to out[wg][words].  Launched through PMR_KLAUNCH like every kernel of the library, so in poison mode every word
// must read PMR_POISON_WORD: tests/test_gpu_poison.py checks the checker.
__global__ __launch_bounds__(256) void k_lds_probe(unsigned *__restrict__ out, unsigned words)
{
    extern __shared__ unsigned pz[];
    for (unsigned i = threadIdx.x; i < words; i += 256u) out[(size_t)blockIdx.x * words + i] = pz[i];
}

extern "C" int pmr_debug_lds_probe(void *d_out, unsigned words, unsigned n_wg)
{
    if (!d_out || !words || words > PMR_POISON_LDS_BYTES / 4u || !n_wg) return (int)hipErrorInvalidValue;
    if (words > 16384u) {                                            // beyond the default 64 KiB dynamic-LDS limit
        static pmr_attr_flags once;
        if (pmr_attr_needed(once)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_probe), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     PMR_POISON_LDS_BYTES);
            if (e != hipSuccess) return (int)e;
        }
    }
    PMR_KLAUNCH(k_lds_probe, dim3(n_wg), dim3(256), (size_t)words * 4u, (hipStream_t)0, (unsigned *)d_out, words);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}
