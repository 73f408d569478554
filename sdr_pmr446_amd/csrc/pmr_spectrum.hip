// pmr_spectrum.hip -- the waterfall line's periodogram on gfx950 (optional).
//
// reference: asgramcf_create(width) + set_scale(-40, 2) src/sdr_pmr446.c:473-477; per block asgramcf_write(resamp_buf, ny) +
// asgramcf_execute :911-912.  liquid's asgram = spgram(4 * width bins, Hann window of `width` samples, one transform every
// width / 2 samples), averaged over the block and reset by execute (oracle/orc_dsp.h states the algorithm).  Because execute
// resets the window buffer too, the transforms of a block depend on that block's samples only:
//   transform t (t = 0 .. ny / delay - 1) = FFT_P( w[i] * x[(t + 1) delay - wlen + i], i < wlen; zero-padded ),  x[< 0] = 0
// so they are independent and read the resampled ring the front end has just written (no copy, no state).
//
//   k_spgram         a workgroup walks a contiguous chunk of transforms: window -> LDS, radix-2 Stockham FFT in LDS (log2 P passes,
//                    natural order in and out), |X|^2 summed per bin in registers; one partial row per workgroup.
//   k_spgram_finish  partial rows summed in a fixed order (deterministic), averaged, clamped at 1e-12, fft-shifted.
// The 10 log10 and the character mapping are host work (pmr_asgram_ascii, pmr_chain.c): P values per block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

typedef float cf __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ cf cmulw(cf a, cf w) { return __builtin_elementwise_fma(cf{a.x, a.x}, w, cf{a.y, a.y} * cf{-w.y, w.x}); }

#define SG_NT 256
#define SG_MAXB 16          /* bins per thread: P <= 4096 */

__global__ __launch_bounds__(SG_NT) void k_spgram(const cf *__restrict__ xr, unsigned long long xr_mask, unsigned long long pos0,
                                                  unsigned ny, unsigned wlen, unsigned log2P, unsigned n_tr, unsigned chunk,
                                                  const float *__restrict__ win, const cf *__restrict__ tw_g,
                                                  float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned P = 1u << log2P, delay = wlen >> 1;
    cf *A = reinterpret_cast<cf *>(smem), *B = A + P, *tw = B + P;          // [P], [P], [P / 2]
    const unsigned tid = threadIdx.x;
    for (unsigned k = tid; k < P / 2; k += SG_NT) tw[k] = tw_g[k];
    float acc[SG_MAXB];
#pragma unroll
    for (int m = 0; m < SG_MAXB; m++) acc[m] = 0.f;
    const unsigned t0 = blockIdx.x * chunk, t1 = min(n_tr, t0 + chunk);
    for (unsigned t = t0; t < t1; t++) {
        __syncthreads();                                                    // twiddles written / previous transform consumed
        const long long s0 = (long long)(t + 1) * delay - (long long)wlen;  // block-relative index of window sample 0
        for (unsigned i = tid; i < P; i += SG_NT) {
            cf v = cf{0.f, 0.f};
            if (i < wlen) {
                const long long s = s0 + i;
                if (s >= 0 && s < (long long)ny) { const cf x = xr[(pos0 + (unsigned long long)s) & xr_mask]; const float w = win[i]; v = cf{x.x * w, x.y * w}; }
            }
            A[i] = v;
        }
        cf *src = A, *dst = B;
        for (unsigned ls = 0; ls < log2P; ls++) {                           // Ns = 1 << ls
            __syncthreads();
            const unsigned Ns = 1u << ls;
            for (unsigned j = tid; j < P / 2; j += SG_NT) {
                const unsigned k = j & (Ns - 1);
                const cf a = src[j], b = cmulw(src[j + P / 2], tw[k << (log2P - 1 - ls)]);   // W_{2 Ns}^k = W_P^{k P / (2 Ns)}
                const unsigned j0 = ((j >> ls) << (ls + 1)) + k;
                dst[j0] = a + b;
                dst[j0 + Ns] = a - b;
            }
            cf *tmp = src; src = dst; dst = tmp;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < SG_MAXB; m++) {
            const unsigned b = tid + SG_NT * m;
            if (b < P) { const cf X = src[b]; acc[m] = fmaf(X.x, X.x, fmaf(X.y, X.y, acc[m])); }
        }
    }
#pragma unroll
    for (int m = 0; m < SG_MAXB; m++) {
        const unsigned b = tid + SG_NT * m;
        if (b < P) partial[(size_t)blockIdx.x * P + b] = acc[m];
    }
}

__global__ __launch_bounds__(SG_NT) void k_spgram_finish(const float *__restrict__ partial, unsigned nwg, unsigned P, unsigned n_tr,
                                                         float *__restrict__ psd_mag)
{
    const unsigned i = blockIdx.x * SG_NT + threadIdx.x;                    // output bin (fft-shifted)
    if (i >= P) return;
    const unsigned b = (i + P / 2) & (P - 1);
    float s = 0.f;
    for (unsigned g = 0; g < nwg; g++) s += partial[(size_t)g * P + b];
    if (s < 1e-12f) s = 1e-12f;
    psd_mag[i] = s * (1.0f / (float)(n_tr ? n_tr : 1u));
}

extern "C" unsigned pmr_spgram_max_workgroups(void) { return 512; }

extern "C" int pmr_launch_spgram(pmr_stream_t s, const void *xr, uint64_t xr_mask, uint64_t pos0, unsigned ny, unsigned wlen,
                                 const float *win, const void *tw, float *partial, float *psd_mag)
{
    unsigned log2P = 0;
    while ((1u << log2P) < 4u * wlen) log2P++;
    const unsigned P = 1u << log2P;
    if (wlen < 8 || P != 4u * wlen || P > SG_NT * SG_MAXB) return (int)hipErrorInvalidValue;
    const unsigned n_tr = ny / (wlen >> 1);
    if (!n_tr) return 0;
    const unsigned maxwg = pmr_spgram_max_workgroups();
    const unsigned chunk = (n_tr + maxwg - 1) / maxwg, nwg = (n_tr + chunk - 1) / chunk;
    const size_t lds = (size_t)(2 * P + P / 2) * sizeof(cf);               /* <= 80 KB at P = 4096 */
    static pmr_attr_flags attr_set{0};
    if (lds > 64 * 1024 && pmr_attr_needed(attr_set))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spgram), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipStream_t st = (hipStream_t)s;
    PMR_KLAUNCH(k_spgram, dim3(nwg), dim3(SG_NT), lds, st, (const cf *)xr, (unsigned long long)xr_mask, (unsigned long long)pos0,
                       ny, wlen, log2P, n_tr, chunk, win, (const cf *)tw, partial);
    PMR_KLAUNCH(k_spgram_finish, dim3((P + SG_NT - 1) / SG_NT), dim3(SG_NT), 0, st, partial, nwg, P, n_tr, psd_mag);
    return (int)hipGetLastError();
}
