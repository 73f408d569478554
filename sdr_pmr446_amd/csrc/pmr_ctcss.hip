// pmr_ctcss.hip -- CTCSS tone detection for all M channels.
//
// reference: complementary low-pass branch  tmp1[k] = delay188(fm[k]) - hp[k]   src/sdr_pmr446.c:884-889
//            ctcss_execute(): dc-block (alpha 5e-4) :606, 38-tone Goertzel bank over CTCSS_BLOCK_SIZE = 2441
//            samples with the decision avg > 120 && max/avg > 10                  :366-409
//
// GPU formulation (everything per channel is linear, so time can be cut into pieces):
//  * the low-pass branch is ONE FIR with taps delta[d-188] - h[d]: the audio FIR kernel (pmr_fir_mfma.hip /
//    k_fir_pair) run a second time on the discriminator ring, writing a time-major ring (done by the host);
//  * the dc-blocker v0 = x - a1 v1, y = v0 - v1 is a first-order linear scan: 256-frame chunks run from zero state
//    (k_ct_dc_agg), a per-channel pass strings the chunk aggregates together (k_ct_dc_scan), k_ct_dc_apply redoes
//    each chunk from its true carry, in place;
//  * the Goertzel recurrence u0' = x + coef u0 - u1 has the impulse response U_n = sin((n+1)w)/sin(w), so after the N
//    samples of a block  u0 = sum_i x_i U_{N-1-i},  u1 = sum_i x_i U_{N-2-i}: a weighted sum that is split over 8 time
//    segments x 38 tones x M channels (k_ct_goertzel) and reduced in a fixed order (k_ct_final), which also carries
//    the partial sums of a block that straddles two calls.  U is tabulated in double on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

#define CT_CHUNK 256

__global__ __launch_bounds__(256) void k_ct_dc_agg(const float *__restrict__ lp, unsigned long long row_mask,
                                                   long long row0, unsigned ns, unsigned M, unsigned log2M, float lam,
                                                   float *__restrict__ agg, unsigned nchunks)
{
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned k = gid & (M - 1), c = gid >> log2M;
    if (c >= nchunks) return;
    const unsigned t0 = c * CT_CHUNK, len = min((unsigned)CT_CHUNK, ns - t0);
    float v = 0.f;
    for (unsigned i = 0; i < len; i++) v = fmaf(lam, v, lp[((unsigned long long)(row0 + t0 + i) & row_mask) * M + k]);
    agg[(size_t)c * M + k] = v;
}

__global__ __launch_bounds__(256) void k_ct_dc_scan(const float *__restrict__ agg, unsigned nchunks, unsigned M,
                                                    float lam_chunk, float lam_last, float *__restrict__ state,
                                                    float *__restrict__ W)
{
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= M) return;
    float v = state[k];
#pragma unroll 8
    for (unsigned c = 0; c < nchunks; c++) {
        W[(size_t)c * M + k] = v;                              // dc-blocker state just before chunk c
        v = fmaf(c + 1 == nchunks ? lam_last : lam_chunk, v, agg[(size_t)c * M + k]);
    }
    state[k] = v;
}

__global__ __launch_bounds__(256) void k_ct_dc_apply(float *__restrict__ lp, unsigned long long row_mask,
                                                     long long row0, unsigned ns, unsigned M, unsigned log2M, float a1,
                                                     const float *__restrict__ W, unsigned nchunks)
{
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned k = gid & (M - 1), c = gid >> log2M;
    if (c >= nchunks) return;
    const unsigned t0 = c * CT_CHUNK, len = min((unsigned)CT_CHUNK, ns - t0);
    float v1 = W[(size_t)c * M + k];
    for (unsigned i = 0; i < len; i++) {
        float *px = lp + ((unsigned long long)(row0 + t0 + i) & row_mask) * M + k;
        const float v0 = __fsub_rn(*px, __fmul_rn(a1, v1));    // iirfilt_rrrf dc blocker, :606
        *px = __fsub_rn(v0, v1);
        v1 = v0;
    }
}

// partial Goertzel sums of one (block, segment): part[(blk*CT_SEG + seg)][k][j][2]
__global__ __launch_bounds__(256) void k_ct_goertzel(const float *__restrict__ lp, unsigned long long row_mask,
                                                     long long row0, unsigned ns, unsigned M, unsigned N,
                                                     const float *__restrict__ U /*[38][N+1], U[j][m+1] = U_m*/,
                                                     float *__restrict__ part, long long b0)
{
    const unsigned blk = blockIdx.x / PMR_CT_SEG, seg = blockIdx.x % PMR_CT_SEG;
    const long long b = b0 + blk;
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;
    long long lo = b * (long long)N + (long long)seg * SL, hi = lo + SL;
    const long long bend = (b + 1) * (long long)N;
    if (hi > bend) hi = bend;
    if (lo < row0) lo = row0;                                  // frames of earlier calls are in the carry
    if (hi > row0 + (long long)ns) hi = row0 + (long long)ns;
    const unsigned items = PMR_CT_TONES * M;
    for (unsigned it = threadIdx.x; it < items; it += 256) {
        const unsigned j = it % PMR_CT_TONES, k = it / PMR_CT_TONES;
        float a0 = 0.f, a1 = 0.f;
        const float *Uj = U + (size_t)j * (N + 1);
        for (long long t = lo; t < hi; t++) {
            const unsigned n = (unsigned)(t - b * (long long)N);
            const float x = lp[((unsigned long long)t & row_mask) * M + k];
            a0 = fmaf(x, Uj[N - n], a0);                       // U_{N-1-n}
            a1 = fmaf(x, Uj[N - 1 - n], a1);                   // U_{N-2-n}
        }
        float *o = part + (((size_t)blk * PMR_CT_SEG + seg) * M + k) * PMR_CT_TONES * 2 + 2 * j;
        o[0] = a0; o[1] = a1;
    }
}

// reduce the segments (fixed order), add the carry of a block begun in an earlier call, decide (:381-406)
__global__ __launch_bounds__(64) void k_ct_final(const float *__restrict__ part, unsigned nblk, unsigned ncomplete,
                                                 unsigned M, const float *__restrict__ coef,
                                                 const float *__restrict__ carry_in, float *__restrict__ carry_out,
                                                 pmr_ctcss_event *__restrict__ events)
{
    const unsigned gid = blockIdx.x * 64u + threadIdx.x;
    const unsigned k = gid % M, blk = gid / M;
    if (blk >= nblk) return;
    float avg = 0.f, maxp = 0.f;
    int maxi = 0;
    const bool complete = blk < ncomplete;
    for (unsigned j = 0; j < PMR_CT_TONES; j++) {
        float u0 = 0.f, u1 = 0.f;
        if (blk == 0) { u0 = carry_in[((size_t)k * PMR_CT_TONES + j) * 2]; u1 = carry_in[((size_t)k * PMR_CT_TONES + j) * 2 + 1]; }
        for (unsigned s = 0; s < PMR_CT_SEG; s++) {
            const float *p = part + (((size_t)blk * PMR_CT_SEG + s) * M + k) * PMR_CT_TONES * 2 + 2 * j;
            u0 += p[0]; u1 += p[1];
        }
        if (complete) {
            const float pw = (u0 * u0) + (u1 * u1) - (coef[j] * u0 * u1);
            avg += pw;
            if (pw > maxp) { maxp = pw; maxi = (int)j; }
        } else {
            carry_out[((size_t)k * PMR_CT_TONES + j) * 2] = u0;
            carry_out[((size_t)k * PMR_CT_TONES + j) * 2 + 1] = u1;
        }
    }
    if (complete) {
        avg /= (float)PMR_CT_TONES;
        pmr_ctcss_event e;
        e.index = maxi; e.detected = (avg > 120.0f) && ((maxp / avg) > 10.0f);
        e.max_power = maxp; e.avg_power = avg;
        events[(size_t)blk * M + k] = e;
    }
}

static inline unsigned ilog2u(unsigned v) { unsigned l = 0; while ((1u << l) < v) l++; return l; }

extern "C" int pmr_launch_ct_dc(pmr_stream_t s, float *lp, uint64_t row_mask, int64_t row0, unsigned ns, unsigned M,
                                float a1, float lam_chunk, float lam_last, float *state, float *agg, float *W)
{
    if (!ns) return 0;
    const unsigned nchunks = (ns + CT_CHUNK - 1) / CT_CHUNK;
    const size_t threads = (size_t)nchunks * M;
    hipLaunchKernelGGL(k_ct_dc_agg, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s, lp,
                       (unsigned long long)row_mask, (long long)row0, ns, M, ilog2u(M), -a1, agg, nchunks);
    hipLaunchKernelGGL(k_ct_dc_scan, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)s, agg, nchunks, M, lam_chunk,
                       lam_last, state, W);
    hipLaunchKernelGGL(k_ct_dc_apply, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s, lp,
                       (unsigned long long)row_mask, (long long)row0, ns, M, ilog2u(M), a1, W, nchunks);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_ct_goertzel(pmr_stream_t s, const float *lp, uint64_t row_mask, int64_t row0, unsigned ns,
                                      unsigned M, unsigned N, const float *U, const float *coef, float *part,
                                      const float *carry_in, float *carry_out, pmr_ctcss_event *events,
                                      unsigned nblk, unsigned ncomplete)
{
    if (!ns || !nblk) return 0;
    const long long b0 = row0 / (long long)N;
    hipLaunchKernelGGL(k_ct_goertzel, dim3(nblk * PMR_CT_SEG), dim3(256), 0, (hipStream_t)s, lp,
                       (unsigned long long)row_mask, (long long)row0, ns, M, N, U, part, b0);
    hipLaunchKernelGGL(k_ct_final, dim3((nblk * M + 63) / 64), dim3(64), 0, (hipStream_t)s, part, nblk, ncomplete, M,
                       coef, carry_in, carry_out, events);
    return (int)hipGetLastError();
}
