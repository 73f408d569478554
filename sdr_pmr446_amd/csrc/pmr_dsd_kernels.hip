// pmr_dsd_kernels.hip -- back end of the `dsd_in` chain (reference src/dsd_in.c:169-175):
// freqdem on the single 12.5 kS/s stream, msresamp_rrrf interpolation to 48 kS/s, int16 conversion.
//
// Everything here runs at <= 48 kS/s per stream, i.e. ~1/80 of the raw sample rate the front end (pmr_frontend.hip)
// digests, so these are plain one-thread-per-output gather kernels over absolute-indexed rings:
//   fm[a]            a = index of the resampled sample           (ring, 13 samples of history needed)
//   u_0[j]           output j of the arbitrary resampler: closed form  T = j*step, input q = T >> 24, bank (T >> 16) & 255
//   u_{g+1}[2i], [2i+1]  half-band interpolator g: delay branch u_g[i-m], filter branch sum_k h1[k] u_g[i-2m+1+k]
// The accumulation order is the portable liquid dot product's: oldest sample first.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

typedef float2 cf;

static __device__ __forceinline__ int16_t dsd_pcm16(float y)
{
    const float s = y * 32767.0f;                 // buf_out_s[i] = out_buf[i] * INT16_MAX, src/dsd_in.c:174
    if (!(s == s)) return 0;
    if (s >= 32767.0f) return 32767;
    if (s <= -32768.0f) return -32768;
    return (int16_t)s;
}

// freqdem_demodulate_block (:169): m = arg(conj(r') r) / (2 pi kf)
__global__ __launch_bounds__(256) void k_dsd_fm(const cf *__restrict__ xr, unsigned long long xr_mask,
                                                unsigned long long a0, unsigned ny, float *__restrict__ fm,
                                                unsigned long long fm_mask, float ref)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ny) return;
    const unsigned long long a = a0 + t;
    const cf cu = xr[a & xr_mask], pv = xr[(a - 1ull) & xr_mask];     // a == 0: the ring is zero there (r' = 0 after reset)
    const float re = fmaf(pv.x, cu.x, pv.y * cu.y);
    const float im = fmaf(pv.x, cu.y, -(pv.y * cu.x));
    fm[a & fm_mask] = pmr_arg(im, re) * ref;
}

// resamp_rrrf_execute inside msresamp_rrrf_execute (:170), outputs j0 .. j0+nu-1
__global__ __launch_bounds__(256) void k_dsd_arb(const float *__restrict__ fm, unsigned long long fm_mask,
                                                 unsigned long long j0, unsigned nu, uint32_t step,
                                                 const float *__restrict__ bank, float *__restrict__ u,
                                                 unsigned long long u_mask, int16_t *__restrict__ pcm,
                                                 float *__restrict__ audio)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nu) return;
    const unsigned long long j = j0 + t, T = j * (unsigned long long)step;
    const unsigned long long q = T >> 24;
    const unsigned idx = (unsigned)(T >> 16) & 255u;
    const float *b = bank + idx * 14u;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 14; k++) acc = fmaf(b[k], fm[(q - 13ull + (unsigned long long)k) & fm_mask], acc);
    if (u) u[j & u_mask] = acc;
    if (pcm) pcm[t] = dsd_pcm16(acc);             // no half-band stage: this is the output
    if (audio) audio[t] = acc;
}

// resamp2_rrrf_interp_execute for inputs i0 .. i0+n-1 of one stage
__global__ __launch_bounds__(256) void k_dsd_hb(const float *__restrict__ in, unsigned long long in_mask,
                                                unsigned long long i0, unsigned n, int m,
                                                const float *__restrict__ h1, float *__restrict__ out,
                                                unsigned long long out_mask, int16_t *__restrict__ pcm,
                                                float *__restrict__ audio)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const unsigned long long i = i0 + t;
    const float y0 = in[(i - (unsigned long long)m) & in_mask];
    float y1 = 0.f;
    for (int k = 0; k < 2 * m; k++)
        y1 = fmaf(h1[k], in[(i - (unsigned long long)(2 * m - 1) + (unsigned long long)k) & in_mask], y1);
    if (out) { out[(2ull * i) & out_mask] = y0; out[(2ull * i + 1ull) & out_mask] = y1; }
    if (pcm) { pcm[2u * t] = dsd_pcm16(y0); pcm[2u * t + 1u] = dsd_pcm16(y1); }
    if (audio) { audio[2u * t] = y0; audio[2u * t + 1u] = y1; }
}

extern "C" int pmr_launch_dsd_fm(pmr_stream_t s, const void *xr, uint64_t xr_mask, uint64_t a0, unsigned ny, float *fm,
                                 uint64_t fm_mask, float ref)
{
    if (!ny) return 0;
    PMR_KLAUNCH(k_dsd_fm, dim3((ny + 255) / 256), dim3(256), 0, (hipStream_t)s, (const cf *)xr,
                       (unsigned long long)xr_mask, (unsigned long long)a0, ny, fm, (unsigned long long)fm_mask, ref);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_dsd_arb(pmr_stream_t s, const float *fm, uint64_t fm_mask, uint64_t j0, unsigned nu,
                                  uint32_t step, const float *bank, float *u, uint64_t u_mask, int16_t *pcm,
                                  float *audio)
{
    if (!nu) return 0;
    PMR_KLAUNCH(k_dsd_arb, dim3((nu + 255) / 256), dim3(256), 0, (hipStream_t)s, fm, (unsigned long long)fm_mask,
                       (unsigned long long)j0, nu, step, bank, u, (unsigned long long)u_mask, pcm, audio);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_dsd_hb(pmr_stream_t s, const float *in, uint64_t in_mask, uint64_t i0, unsigned n, int m,
                                 const float *h1, float *out, uint64_t out_mask, int16_t *pcm, float *audio)
{
    if (!n) return 0;
    PMR_KLAUNCH(k_dsd_hb, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, in, (unsigned long long)in_mask,
                       (unsigned long long)i0, n, m, h1, out, (unsigned long long)out_mask, pcm, audio);
    return (int)hipGetLastError();
}
