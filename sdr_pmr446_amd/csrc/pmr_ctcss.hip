// pmr_ctcss.hip -- CTCSS tone detection for all M channels.
//
// reference: complementary low-pass branch  tmp1[k] = delay188(fm[k]) - hp[k]   src/sdr_pmr446.c:884-889
//            ctcss_execute(): dc-block (alpha 5e-4) :606, 38-tone Goertzel bank over CTCSS_BLOCK_SIZE = 2441
//            samples with the decision avg > 120 && max/avg > 10                  :366-409
//
// GPU formulation (everything per channel is linear, so time can be cut into pieces):
//  * the low-pass branch is ONE FIR with taps delta[d-188] - h[d]: the audio FIR kernel (pmr_fir_mfma.hip /
//    k_fir_pair) run a second time on the discriminator ring, writing a time-major ring (done by the host);
//  * the dc-blocker v0 = x - a1 v1, y = v0 - v1 is a first-order linear scan: 64-frame chunks run from zero state
//    (k_ct_dc_agg), a workgroup per channel strings the chunk aggregates together with a parallel scan of affine maps
//    (k_ct_dc_scan), k_ct_dc_apply redoes each chunk from its true carry, in place;
//  * the Goertzel recurrence u0' = x + coef u0 - u1 has the impulse response U_n = sin((n+1)w)/sin(w), so after the N
//    samples of a block  u0 = sum_i x_i U_{N-1-i},  u1 = sum_i x_i U_{N-2-i}: a weighted sum that is split over 16 time
//    segments x 38 tones x M channels (k_ct_goertzel: LDS-tiled [16 channels] x [frames] . [frames] x [38 tones]) and reduced in a fixed order (k_ct_final), which also carries
//    the partial sums of a block that straddles two calls.  U is tabulated in double on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

#define CT_CHUNK PMR_CT_CHUNK            /* frames per dc-scan chunk */

// chunk aggregates of the dc blocker from zero state: agg[c][k] = sum_i lam^(len-1-i) x[t0 + i][k].  The loads of a batch are all
// issued before the dependent fma chain consumes them (a run-time-bounded loop pays the L2 latency once per sample).
__global__ __launch_bounds__(256) void k_ct_dc_agg(const float *__restrict__ lp, unsigned long long row_mask,
                                                   long long row0, unsigned ns, unsigned M, unsigned log2M, float lam,
                                                   float *__restrict__ agg, unsigned nchunks)
{
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned k = gid & (M - 1), c = gid >> log2M;
    if (c >= nchunks) return;
    const unsigned t0 = c * CT_CHUNK, len = min((unsigned)CT_CHUNK, ns - t0);
    float v = 0.f;
    for (unsigned i0 = 0; i0 < len; i0 += 16) {
        float x[16];
#pragma unroll
        for (unsigned u = 0; u < 16; u++)
            x[u] = i0 + u < len ? lp[((unsigned long long)(row0 + t0 + i0 + u) & row_mask) * M + k] : 0.f;
#pragma unroll
        for (unsigned u = 0; u < 16; u++) if (i0 + u < len) v = fmaf(lam, v, x[u]);
    }
    agg[(size_t)c * M + k] = v;
}

// dc-blocker state just before every chunk: a first-order linear scan over the chunk aggregates, one WORKGROUP per channel
// (a thread walks a contiguous run of chunks; the runs are strung together by a Hillis-Steele scan of (multiplier, value) pairs)
__global__ __launch_bounds__(256) void k_ct_dc_scan(const float *__restrict__ agg, unsigned nchunks, unsigned M,
                                                    float lam_chunk, float lam_last, float *__restrict__ state,
                                                    float *__restrict__ W)
{
    __shared__ float sP[256], sA[256];
    const unsigned k = blockIdx.x, t = threadIdx.x;
    const unsigned per = (nchunks + 255u) / 256u, c0 = t * per, c1 = min(nchunks, c0 + per);
    float P = 1.f, A = 0.f;                                    // run from zero state: v_out = P v_in + A
    for (unsigned c = c0; c < c1; c++) {
        const float l = c + 1 == nchunks ? lam_last : lam_chunk;
        A = fmaf(l, A, agg[(size_t)c * M + k]);
        P *= l;
    }
    sP[t] = P; sA[t] = A;
    __syncthreads();
#pragma unroll
    for (unsigned d = 1; d < 256; d <<= 1) {                   // inclusive scan of the affine maps
        float p2 = 1.f, a2 = 0.f;
        if (t >= d) { p2 = sP[t - d]; a2 = sA[t - d]; }
        __syncthreads();
        if (t >= d) { sA[t] = fmaf(sP[t], a2, sA[t]); sP[t] = sP[t] * p2; }
        __syncthreads();
    }
    const float s0 = state[k];
    float v = t == 0 ? s0 : fmaf(sP[t - 1], s0, sA[t - 1]);    // state before this thread's run
    for (unsigned c = c0; c < c1; c++) {
        W[(size_t)c * M + k] = v;
        v = fmaf(c + 1 == nchunks ? lam_last : lam_chunk, v, agg[(size_t)c * M + k]);
    }
    if (t == 255) state[k] = fmaf(sP[255], s0, sA[255]);
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
    for (unsigned i0 = 0; i0 < len; i0 += 16) {
        float x[16];
#pragma unroll
        for (unsigned u = 0; u < 16; u++)
            x[u] = i0 + u < len ? lp[((unsigned long long)(row0 + t0 + i0 + u) & row_mask) * M + k] : 0.f;
#pragma unroll
        for (unsigned u = 0; u < 16; u++) {
            if (i0 + u < len) {
                const float v0 = __fsub_rn(x[u], __fmul_rn(a1, v1));    // iirfilt_rrrf dc blocker, :606
                lp[((unsigned long long)(row0 + t0 + i0 + u) & row_mask) * M + k] = __fsub_rn(v0, v1);
                v1 = v0;
            }
        }
    }
}

// partial Goertzel sums of one (block, segment, group of 16 channels): part[(blk*CT_SEG + seg)][k][j][2].
// A block is the product  [channels] x [frames] . [frames] x [38 tones x 2]  with the weights U: the segment's samples (16
// channels) and the matching window of U (38 tones) are staged in LDS once; a thread owns one tone and FOUR channels, so an
// iteration is one ds_read_b128 of samples (broadcast among the threads of a quad) + one weight for eight FMAs.
#define CG_T 192                                                   /* threads: 38 tones x 4 channel quads = 152 active */
__global__ __launch_bounds__(CG_T) void k_ct_goertzel(const float *__restrict__ lp, unsigned long long row_mask,
                                                     long long row0, unsigned ns, unsigned M, unsigned N,
                                                     const float *__restrict__ U /*[38][N+1], U[j][m+1] = U_m*/,
                                                     float *__restrict__ part, long long b0)
{
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;         // frames per segment
    float *xs = reinterpret_cast<float *>(smem_c);                 // [SL][16]
    float *us = xs + (size_t)SL * 16;                              // [38][SL + 2] (odd-ish stride: conflict-free over the tones)
    const unsigned US = SL + 2 + ((SL & 1) ? 0 : 1);               // row stride of us, odd
    const unsigned tid = threadIdx.x;
    const unsigned bs = blockIdx.x, blk = bs / PMR_CT_SEG, seg = bs % PMR_CT_SEG, cg = blockIdx.y * 16u;
    const long long b = b0 + blk;
    long long lo = b * (long long)N + (long long)seg * SL, hi = lo + SL;
    const long long bend = (b + 1) * (long long)N;
    if (hi > bend) hi = bend;
    if (lo < row0) lo = row0;                                      // frames of earlier calls are in the carry
    if (hi > row0 + (long long)ns) hi = row0 + (long long)ns;
    const int len = hi > lo ? (int)(hi - lo) : 0;
    const unsigned n0 = len ? (unsigned)(lo - b * (long long)N) : 0;   // position of the first sample inside the block
    for (int i = tid; i < len * 4; i += CG_T) {                    // samples: 64 bytes per frame
        const int r = i >> 2, q4 = (i & 3) * 4;
        const float *src = lp + ((unsigned long long)(lo + r) & row_mask) * M + cg + q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cg + q4 + 4 <= M) v = *reinterpret_cast<const float4 *>(src);                 // M >= 4 is a power of two
        else if (cg + q4 < M) { v.x = src[0]; if (cg + q4 + 1 < M) v.y = src[1]; }           // M = 2
        *reinterpret_cast<float4 *>(xs + r * 16 + q4) = v;
    }
    // weights: sample n of the block meets U[N - n] (-> u0) and U[N - 1 - n] (-> u1); window = U[N - n0 - len .. N - n0]
    for (int i = tid; i < (int)PMR_CT_TONES * (len + 1); i += CG_T) {
        const int j = i / (len + 1), m = i % (len + 1);
        us[j * US + m] = U[(size_t)j * (N + 1) + (N - n0 - len) + m];
    }
    __syncthreads();
    const unsigned j = tid >> 2, kq = tid & 3u;
    if (j >= PMR_CT_TONES) return;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    const float *uj = us + j * US;
    if (len) {
        float u_hi = uj[len];                                      // U[N - n0]: weight of the first sample in u0
        for (int r = 0; r < len; r++) {
            const float u_lo = uj[len - 1 - r];                    // U[N - 1 - n]
            const float4 x = *reinterpret_cast<const float4 *>(xs + r * 16 + 4 * kq);
            a0[0] = fmaf(x.x, u_hi, a0[0]); a0[1] = fmaf(x.y, u_hi, a0[1]); a0[2] = fmaf(x.z, u_hi, a0[2]); a0[3] = fmaf(x.w, u_hi, a0[3]);
            a1[0] = fmaf(x.x, u_lo, a1[0]); a1[1] = fmaf(x.y, u_lo, a1[1]); a1[2] = fmaf(x.z, u_lo, a1[2]); a1[3] = fmaf(x.w, u_lo, a1[3]);
            u_hi = u_lo;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const unsigned k = cg + 4 * kq + q;
        if (k >= M) break;
        float *o = part + (((size_t)blk * PMR_CT_SEG + seg) * M + k) * PMR_CT_TONES * 2 + 2 * j;
        o[0] = a0[q]; o[1] = a1[q];
    }
}

// reduce the segments (fixed order), add the carry of a block begun in an earlier call, decide (:381-406).
// One 64-thread workgroup per (block, channel): lane j < 38 sums tone j over the segments, lane 0 takes the decision.
__global__ __launch_bounds__(64) void k_ct_final(const float *__restrict__ part, unsigned nblk, unsigned ncomplete,
                                                 unsigned M, const float *__restrict__ coef,
                                                 const float *__restrict__ carry_in, float *__restrict__ carry_out,
                                                 pmr_ctcss_event *__restrict__ events)
{
    __shared__ float spw[PMR_CT_TONES];
    const unsigned k = blockIdx.x % M, blk = blockIdx.x / M, j = threadIdx.x;
    const bool complete = blk < ncomplete;
    if (j < PMR_CT_TONES) {
        float u0 = 0.f, u1 = 0.f;
        if (blk == 0) { u0 = carry_in[((size_t)k * PMR_CT_TONES + j) * 2]; u1 = carry_in[((size_t)k * PMR_CT_TONES + j) * 2 + 1]; }
        float2 pv[PMR_CT_SEG];
#pragma unroll
        for (unsigned s = 0; s < PMR_CT_SEG; s++)
            pv[s] = *reinterpret_cast<const float2 *>(part + (((size_t)blk * PMR_CT_SEG + s) * M + k) * PMR_CT_TONES * 2 + 2 * j);
#pragma unroll
        for (unsigned s = 0; s < PMR_CT_SEG; s++) { u0 += pv[s].x; u1 += pv[s].y; }
        if (complete) spw[j] = (u0 * u0) + (u1 * u1) - (coef[j] * u0 * u1);
        else {
            carry_out[((size_t)k * PMR_CT_TONES + j) * 2] = u0;
            carry_out[((size_t)k * PMR_CT_TONES + j) * 2 + 1] = u1;
        }
    }
    __syncthreads();
    if (complete && j == 0) {
        float avg = 0.f, maxp = 0.f;
        int maxi = 0;
        for (unsigned t = 0; t < PMR_CT_TONES; t++) {
            const float pw = spw[t];
            avg += pw;
            if (pw > maxp) { maxp = pw; maxi = (int)t; }
        }
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
    hipLaunchKernelGGL(k_ct_dc_scan, dim3(M), dim3(256), 0, (hipStream_t)s, agg, nchunks, M, lam_chunk, lam_last, state, W);
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
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG, US = SL + 2 + ((SL & 1) ? 0 : 1);
    const size_t lds = ((size_t)SL * 16 + (size_t)PMR_CT_TONES * US) * sizeof(float);
    hipLaunchKernelGGL(k_ct_goertzel, dim3(nblk * PMR_CT_SEG, (M + 15) / 16), dim3(CG_T), lds, (hipStream_t)s, lp,
                       (unsigned long long)row_mask, (long long)row0, ns, M, N, U, part, b0);
    hipLaunchKernelGGL(k_ct_final, dim3(nblk * M), dim3(64), 0, (hipStream_t)s, part, nblk, ncomplete, M,
                       coef, carry_in, carry_out, events);
    return (int)hipGetLastError();
}
