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

#include "pmr_kernels.h"

#define PMR_POISON_WORD 0x7FA0DEADu

static int g_poison = -1;                                          // -1: not read yet (environment PMR_DEBUG_POISON)

extern "C" int pmr_debug_poison_enabled(void)
{
    if (g_poison < 0) { const char *e = getenv("PMR_DEBUG_POISON"); g_poison = (e && e[0] && e[0] != '0') ? 1 : 0; }
    return g_poison;
}

extern "C" int pmr_debug_poison(int on)
{
    const int was = pmr_debug_poison_enabled();
    g_poison = on ? 1 : 0;
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

extern "C" int pmr_debug_poison_lds(pmr_stream_t s)
{
    static int dev_cus[64], dev_lds[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!dev_cus[dev]) {
        int cus = 0, lds = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        if (cus <= 0) cus = 256;
        if (lds <= 0 || lds > 160 * 1024) lds = 160 * 1024;        // gfx950: 160 KiB per CU, all of it allocatable by one workgroup
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        dev_cus[dev] = cus; dev_lds[dev] = lds;
    }
    const int lds = dev_lds[dev], per_cu = (160 * 1024) / lds;      // 1 on gfx950
    hipLaunchKernelGGL(k_poison_lds, dim3((unsigned)(dev_cus[dev] * (per_cu > 0 ? per_cu : 1))), dim3(1024), (size_t)lds, (hipStream_t)s,
                       (unsigned)lds / 4u, PMR_POISON_WORD);
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
    if (!d_out || !words || words > 16384u || !n_wg) return (int)hipErrorInvalidValue;
    PMR_KLAUNCH(k_lds_probe, dim3(n_wg), dim3(256), (size_t)words * 4u, (hipStream_t)0, (unsigned *)d_out, words);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}
