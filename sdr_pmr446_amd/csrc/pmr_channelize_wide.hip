// pmr_channelize_wide.hip -- channelizer + discriminator for WIDE banks (M = 64, 256, 1024, 4096: powers of four) on gfx950.
//
// reference: NCO shift src/sdr_pmr446.c:808-812, firpfbch_crcf_analyzer_execute :814, transpose :819-821, freqdem :881.
// Sums: tests/chain_model.py  (X_c[t] = sum_k taps_t[k][c] * xm[(t - (p-1) + k) * M + c],  y[t] = FFT_M(X[t])).
//
// A block holds few frames when M is large (cfg5: 838 frames of 1024 channels), so a frame-tiled kernel yields a few hundred
// workgroups that each walk filter bank -> log2(M) barrier passes of a radix-2 FFT -> discriminator: latency, not work, set its
// time.  Here the two halves get the parallelism each one has:
//   k_pfb_wide   filter bank only, parallel over CHANNELS x frame groups: thread = (channel, F = 8 consecutive frames); the 26
//                branch taps sit in registers, the group's F + 25 input rows come straight from the resampled ring (a wave
//                reads 512 contiguous bytes per row), every sample is loaded and NCO-mixed once and feeds up to F frames.
//                X goes to a scratch array [ns + 1][M] (row 0 = the frame before the block, recomputed from the ring so calls
//                stay independent); it is 8 * rate bytes per input sample and lives in L2 / Infinity Cache.
//   k_fft_disc   FFT + discriminator, parallel over FRAMES: a workgroup owns FPW consecutive rows of X (the first is the
//                previous frame), runs FPW radix-4 Stockham FFTs side by side in LDS (log4(M) passes instead of log2(M),
//                natural order in and out, twiddles from an LDS copy of the table) and writes the FPW - 1 new
//                discriminator rows, the channel-major tap-off and the RSSI partial sums.
// The FFT's butterfly order differs from the oracle's radix-2 DIT; the difference is float32 rounding (~1e-7 of the frame's
// RMS), far inside the 1e-5 / +-1 LSB parity tolerances (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"
#include "pmr_carry_load.hpp"

typedef float cf __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ cf cfm(float r, float i) { return cf{r, i}; }
static __device__ __forceinline__ cf cmul(cf a, cf w) { return __builtin_elementwise_fma(cf{a.x, a.x}, w, cf{a.y, a.y} * cf{-w.y, w.x}); }

#ifndef PW_F
#define PW_F 8          /* frames per (channel, group) work item of the filter bank */
#endif
#ifndef PW_FPW1024
#define PW_FPW1024 2    /* rows of X per k_fft_disc workgroup at M = 1024 (the first is the previous frame, transformed again) */
#endif
#define PW_P 26         /* branch taps (2 m, m = 13: reference :437) */

#ifndef PW_MINB
#define PW_MINB 4       /* workgroups per CU the register budget allows for: 4 -> <= 128 VGPRs, a wave fits beside four front-end waves on a SIMD */
#endif
#ifndef PW_RB
#define PW_RB 9         /* rows per batch: loads first, then the MACs */
#endif
__global__ __launch_bounds__(256, PW_MINB) void k_pfb_wide(pmr_chan_params q, unsigned log2M, cf *__restrict__ Xg)
{
    const unsigned M = q.M, nrows = q.ns + 1;                    // row r of Xg <-> frame (frame0 - 1 + r)
    const unsigned w = pmr_xcd_contiguous(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
    const unsigned c = w & (M - 1), r0 = (w >> log2M) * PW_F;
    if (r0 >= nrows) return;
    const cf *__restrict__ xr = (const cf *)q.xr;
    const cf *__restrict__ nco_cs = (const cf *)q.nco_cs;
    const unsigned nco_mask = q.nco_period - 1, xr_mask32 = (unsigned)q.xr_mask;
    float h[PW_P];
#pragma unroll
    for (int k = 0; k < PW_P; k++) h[k] = q.taps_t[k * M + c];
    cf acc[PW_F];
#pragma unroll
    for (int f = 0; f < PW_F; f++) acc[f] = cfm(0.f, 0.f);
    // low 32 bits of the absolute sample index are all the ring / NCO masks need (indices before the stream start wrap into
    // the zero-initialised top of the ring, as in k_channelize)
    const long long fbase = (long long)q.frame0 - 1 + r0 - (PW_P - 1);      // absolute frame of (f = 0, k = 0)
    const unsigned a0 = (unsigned)((unsigned long long)fbase * (unsigned long long)M) + c;
    // the NCO table has period 2 M (reference :432-434: d theta = -2 pi (M-1)/(2M)), so a thread meets only two factors: one on
    // even rows of its window, one on odd rows
    const cf cs_e = nco_cs[a0 & nco_mask], cs_o = nco_cs[(a0 + M) & nco_mask];
    constexpr int RB = PW_RB;
#pragma unroll
    for (int rr0 = 0; rr0 < PW_F + PW_P - 1; rr0 += RB) {
        cf xm[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const int r = rr0 + u;
            if (r < PW_F + PW_P - 1) {
                const unsigned a = a0 + (unsigned)r * M;
                const cf x = xr[a & xr_mask32];
                const cf cs = (r & 1) ? cs_o : cs_e;
                xm[u] = cfm(fmaf(x.x, cs.x, x.y * cs.y), fmaf(x.y, cs.x, -(x.x * cs.y)));   // x * conj(e^{j theta})
            }
        }
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const int r = rr0 + u;
            if (r < PW_F + PW_P - 1) {
#pragma unroll
                for (int f = (r - PW_P + 1 > 0 ? r - PW_P + 1 : 0); f <= (r < PW_F - 1 ? r : PW_F - 1); f++)
                    acc[f] = __builtin_elementwise_fma(cf{h[r - f], h[r - f]}, xm[u], acc[f]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < PW_F; f++)
        if (r0 + f < nrows) Xg[(size_t)(r0 + f) * M + c] = acc[f];
}

// M-point forward FFTs of FPW rows, radix-4 Stockham (natural order in and out), IN PLACE in one LDS array: a pass reads its
// butterflies' inputs into registers, the workgroup synchronises, then the outputs go back to the same array (two barriers per
// pass, half the LDS of a ping-pong pair: 20 KB at M = 1024, so the kernel co-resides with the front end's tiles).
//   pass Ns = 1, 4, 16, ...:  butterfly j of an FFT reads x[j + r M/4] (r = 0..3), multiplies by W_{4 Ns}^{r (j mod Ns)}, takes the
//   4-point DFT and writes x[(j / Ns) 4 Ns + (j mod Ns) + r Ns].
template <int M, int FPW>
__global__ __launch_bounds__(256) void k_fft_disc(pmr_chan_params q, const cf *__restrict__ Xg)
{
    constexpr int NB = FPW * (M / 4), NPT = NB / 256;            // butterflies per pass; per thread
    static_assert(NB % 256 == 0 && NPT >= 1, "butterflies must tile the workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *A = reinterpret_cast<cf *>(smem), *tw = A + FPW * M;                        // [FPW][M], [M/2]
    const int tid = threadIdx.x;
    const unsigned ns = q.ns;
    const unsigned wg = pmr_xcd_contiguous(blockIdx.x, gridDim.x);
    const unsigned row0 = wg * (FPW - 1);                        // first row of Xg this workgroup reads (= previous frame)
    const unsigned nrow = min((unsigned)FPW, ns + 1 - row0);     // valid rows; new frames: nrow - 1

    for (int k = tid; k < M / 2; k += 256) tw[k] = ((const cf *)q.fft_tw)[k];
    {
        const float4 *src = reinterpret_cast<const float4 *>(Xg + (size_t)row0 * M);
        float4 *dst = reinterpret_cast<float4 *>(A);
        for (int i = tid; i < FPW * M / 2; i += 256) dst[i] = (unsigned)(2 * i) < nrow * M ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const auto twid = [&](int i) {                                // e^{-2 pi j i / M}, i in [0, M)
        const cf w = tw[i & (M / 2 - 1)];
        return i >= M / 2 ? -w : w;
    };
#pragma unroll
    for (int Ns = 1; Ns < M; Ns *= 4) {
        cf y[NPT][4];
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            const int idx = tid + 256 * u, f = idx / (M / 4), j = idx % (M / 4), k = j & (Ns - 1);
            const cf *x = A + f * M + j;
            cf v0 = x[0], v1 = x[M / 4], v2 = x[M / 2], v3 = x[3 * M / 4];
            if (Ns > 1) {
                const int ti = k * (M / (4 * Ns));
                v1 = cmul(v1, twid(ti)); v2 = cmul(v2, twid(2 * ti)); v3 = cmul(v3, twid(3 * ti));
            }
            const cf t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, d = v1 - v3, t3 = cfm(d.y, -d.x);     // t3 = -j (v1 - v3)
            y[u][0] = t0 + t2; y[u][1] = t1 + t3; y[u][2] = t0 - t2; y[u][3] = t1 - t3;
        }
        __syncthreads();                                          // every butterfly of the pass holds its inputs
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            const int idx = tid + 256 * u, f = idx / (M / 4), j = idx % (M / 4), k = j & (Ns - 1);
            cf *o = A + f * M + (j - k) * 4 + k;
            o[0] = y[u][0]; o[Ns] = y[u][1]; o[2 * Ns] = y[u][2]; o[3 * Ns] = y[u][3];
        }
        __syncthreads();
    }
    // ---- discriminator (:881) m = arg(conj(prev) cur) / (2 pi kf) for the new frames, tap-off, RSSI partial sums ----
    const cf *Y = A;
    cf *__restrict__ chan_out = (cf *)q.chan_out;
    const unsigned nnew = nrow - 1;
    // compile-time trip count (guarded): the arg() chains of a thread's elements are independent and interleave -- as a loop
    // bounded by nnew * M they ran one after the other
    constexpr unsigned NW = ((FPW - 1) * M + 255) / 256;
#pragma unroll
    for (unsigned u = 0; u < NW; u++) {
        const unsigned wi = tid + 256u * u;
        if (wi < nnew * M) {
            const unsigned f = wi / M, k = wi % M;
            const cf pv = Y[f * M + k], cu = Y[(f + 1) * M + k];
            const float re = fmaf(pv.x, cu.x, pv.y * cu.y), im = fmaf(pv.x, cu.y, -(pv.y * cu.x));
            const unsigned t = row0 + f;                         // frame relative to frame0
            q.fm[((unsigned long long)(q.frame0 + t) & q.fm_mask & PMR_EXP_ROW_AND) * M + k] = pmr_arg(im, re) * q.fm_ref;
            if (chan_out) chan_out[(size_t)k * q.chan_stride + t] = cu;
        }
    }
    if (row0 == 0 && q.reset_flags && nnew) {                    // freqdem_reset: previous sample = 0 -> arg(0) = 0, flagged channels'
        for (unsigned k = tid; k < M; k += 256)                  //  first output of the call (element (0, k) was written by thread k % 256)
            if (q.reset_flags[k]) q.fm[((unsigned long long)q.frame0 & q.fm_mask) * M + k] = 0.f;
    }
    if (q.rssi_part) {
        for (unsigned k = tid; k < M; k += 256) {
            float a = 0.f;
            for (unsigned f = 0; f < nnew; f++) { const cf cu = Y[(f + 1) * M + k]; a += hypotf(cu.x, cu.y); }
            q.rssi_part[(size_t)wg * M + k] = a;
        }
    }
}

// M = 256: filter bank, FFT and discriminator in ONE kernel -- a workgroup is the 256 channels of G = 8 consecutive new frames
// (+ the frame before them, recomputed, for the discriminator): thread = channel, the nine bank outputs go to LDS instead of the
// scratch array, the nine 256-point FFTs run there (same radix-4 passes as k_fft_disc), then the discriminator.  Saves the
// scratch array's round trip (8 * rate B per input sample written and read: 54 MB per 2^26-sample block at cfg3) and a kernel
// boundary; the arithmetic is that of k_pfb_wide + k_fft_disc, operation for operation.
// FIX: the front end's dc carry is subtracted from the samples as they are loaded (pmr_carry_fix / pmr_carry_load.hpp).  A
// thread's rows are 256 outputs apart = M * step / 2^24 ~ 307 decimated samples at cfg3, more than one front-end tile (217) and
// less than two: NOV = 2 compare-and-subtract steps per row.
// Frames per workgroup (round 4): 24 new frames (+ the previous one) for blocks of thousands of frames, 8 for short ones.  A thread
// loads, corrects and mixes (G + 26) / G rows per frame: 4.25 at G = 8, 2.1 at G = 24.  Round 3 measured G = 16 10 % faster ALONE
// and 4 % slower IN THE CHAIN (its 35 KB of LDS beside the direct audio FIR's 36 KB); with the FFT form of the FIR (8.7 KB) the
// balance turned: cfg3 chain 430.8 / 431.7 (G = 8), 436.0 / 436.7 (12), 441.4 / 441.1 (16), 440.8 / 440.7 (20), 445.1 / 446.8 GS/s (24)
// on one box (profiles/r04_ab_log.txt r4q).  Beyond 24 the accumulators no longer fit 128 VGPRs (spills).  RB = rows per load
// batch (two batches in flight): sized so that nothing spills.
#ifndef PF_G
#define PF_G 24
#endif
#ifndef PF_RB
#define PF_RB 3
#endif
#ifndef PF_MINB
#define PF_MINB 4          /* workgroups per CU the register budget allows for (4 -> <= 128 VGPRs) */
#endif
#define PF_G_SMALL 8        /* blocks of < PF_SMALL_NS frames (the synchronous small-block calls): more, shorter workgroups */
#define PF_RB_SMALL 9
#define PF_SMALL_NS 2048u
template <bool FIX, int G, int RBATCH>
__global__ __launch_bounds__(256, PF_MINB) void k_channelize_fused256(pmr_chan_params q)
{
    constexpr int M = 256, NF = G + 1, NROW = NF + PW_P - 1;                // frames per workgroup (first = previous); input rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *A = reinterpret_cast<cf *>(smem), *tw = A + NF * M;                 // [NF][M], [M/2]
    const int tid = threadIdx.x;
    const unsigned ns = q.ns;
    const unsigned wg = pmr_xcd_contiguous(blockIdx.x, gridDim.x);
    const unsigned R0 = wg * G;                                             // row R0 <-> frame frame0 - 1 + R0 (the previous frame)
    if (tid < M / 2) tw[tid] = ((const cf *)q.fft_tw)[tid];
    pmr_carry_lds ct;
    if constexpr (FIX) {
        ct = pmr_carry_setup<256>(q.fix, reinterpret_cast<float *>(tw + M / 2),
                                  ((long long)q.frame0 - 1 + R0 - (PW_P - 1)) * M - (long long)q.fix.pos0, tid);
        __syncthreads();
    }
    // ---- filter bank (k_pfb_wide's arithmetic), thread = channel ----
    {
        const unsigned c = (unsigned)tid;
        const cf *__restrict__ xr = (const cf *)q.xr;
        const cf *__restrict__ nco_cs = (const cf *)q.nco_cs;
        const unsigned nco_mask = q.nco_period - 1, xr_mask32 = (unsigned)q.xr_mask;
        float h[PW_P];
#pragma unroll
        for (int k = 0; k < PW_P; k++) h[k] = q.taps_t[k * M + c];
        cf acc[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) acc[f] = cfm(0.f, 0.f);
        const long long fbase = (long long)q.frame0 - 1 + R0 - (PW_P - 1);
        const unsigned a0 = (unsigned)((unsigned long long)fbase * (unsigned long long)M) + c;
        const cf cs_e = nco_cs[a0 & nco_mask], cs_o = nco_cs[(a0 + M) & nco_mask];
        pmr_carry_state cst;
        const unsigned long long dph = (unsigned long long)M * q.fix.step;
        if constexpr (FIX) cst = pmr_carry_init(q.fix, ct, fbase * (long long)M + (long long)c - (long long)q.fix.pos0);
        // rows in batches of RB, SOFTWARE-PIPELINED: batch b + 1 is requested before batch b is consumed, so only the first batch's
        // L2 / HBM latency is exposed (round 3 loaded, waited and computed batch by batch).  Measured neutral (26 us alone either
        // way, profiles/r04_ab_log.txt r4g: the bank phase is 19.5 of the kernel's 26 us, 6.6 of them the carry arithmetic)
        constexpr int RB = RBATCH, NBATCH = (NROW + RB - 1) / RB;
        cf xq[2][RB];
        const auto request = [&](int b, cf (&dst)[RB]) {
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int r = b * RB + u;
                if (r < NROW) dst[u] = xr[(a0 + (unsigned)r * M) & xr_mask32];
            }
        };
        request(0, xq[0]);
#pragma unroll
        for (int b = 0; b < NBATCH; b++) {
            cf (&xm)[RB] = xq[b & 1];
            if (b + 1 < NBATCH) request(b + 1, xq[(b + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);            // the next batch's loads are issued ahead of everything that consumes this one
            const int rr0 = b * RB;
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int r = rr0 + u;
                if (r < NROW) {
                    cf x = xm[u];
                    if constexpr (FIX) x = pmr_carry_apply<2>(q.fix, ct, cst, x, M, dph);
                    const cf cs = (r & 1) ? cs_o : cs_e;
                    xm[u] = cfm(fmaf(x.x, cs.x, x.y * cs.y), fmaf(x.y, cs.x, -(x.x * cs.y)));
#pragma unroll
                    for (int f = (r - PW_P + 1 > 0 ? r - PW_P + 1 : 0); f <= (r < NF - 1 ? r : NF - 1); f++)
                        acc[f] = __builtin_elementwise_fma(cf{h[r - f], h[r - f]}, xm[u], acc[f]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < NF; f++) A[f * M + c] = acc[f];
    }
    __syncthreads();
    // ---- NF forward FFTs of M points, radix-4 Stockham in place (k_fft_disc's passes) ----
    const auto twid = [&](int i) { const cf w = tw[i & (M / 2 - 1)]; return i >= M / 2 ? -w : w; };
    constexpr int NB = NF * (M / 4), NPT = (NB + 255) / 256;
#pragma unroll
    for (int Ns = 1; Ns < M; Ns *= 4) {
        cf y[NPT][4];
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            const int idx = tid + 256 * u;
            if (idx < NB) {
                const int f = idx / (M / 4), j = idx % (M / 4), k = j & (Ns - 1);
                const cf *x = A + f * M + j;
                cf v0 = x[0], v1 = x[M / 4], v2 = x[M / 2], v3 = x[3 * M / 4];
                if (Ns > 1) {
                    const int ti = k * (M / (4 * Ns));
                    v1 = cmul(v1, twid(ti)); v2 = cmul(v2, twid(2 * ti)); v3 = cmul(v3, twid(3 * ti));
                }
                const cf t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, d = v1 - v3, t3 = cfm(d.y, -d.x);
                y[u][0] = t0 + t2; y[u][1] = t1 + t3; y[u][2] = t0 - t2; y[u][3] = t1 - t3;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            const int idx = tid + 256 * u;
            if (idx < NB) {
                const int f = idx / (M / 4), j = idx % (M / 4), k = j & (Ns - 1);
                cf *o = A + f * M + (j - k) * 4 + k;
                o[0] = y[u][0]; o[Ns] = y[u][1]; o[2 * Ns] = y[u][2]; o[3 * Ns] = y[u][3];
            }
        }
        __syncthreads();
    }
    // ---- discriminator, tap-off, RSSI partial sums (k_fft_disc's epilogue) ----
    cf *__restrict__ chan_out = (cf *)q.chan_out;
    const unsigned nnew = R0 >= ns ? 0u : min((unsigned)G, ns - R0);
    {
        // thread = channel: the frame before is the previous iteration's `cu`; fixed trip count, so the eight arg() chains interleave
        const unsigned k = (unsigned)tid;
        cf pv = A[k];
#pragma unroll
        for (unsigned f = 0; f < (unsigned)G; f++) {
            const cf cu = A[(f + 1) * M + k];
            if (f < nnew) {
                const float re = fmaf(pv.x, cu.x, pv.y * cu.y), im = fmaf(pv.x, cu.y, -(pv.y * cu.x));
                const unsigned t = R0 + f;
                q.fm[((unsigned long long)(q.frame0 + t) & q.fm_mask & PMR_EXP_ROW_AND) * M + k] = pmr_arg(im, re) * q.fm_ref;
                if (chan_out) chan_out[(size_t)k * q.chan_stride + t] = cu;
            }
            pv = cu;
        }
    }
    if (R0 == 0 && q.reset_flags && nnew) {              // freqdem_reset: arg(0) = 0 for the flagged channels' first output of the call;
        const unsigned k = (unsigned)tid;                //  thread k wrote (frame 0, channel k) above (M = 256 threads)
        if (q.reset_flags[k]) q.fm[((unsigned long long)q.frame0 & q.fm_mask) * M + k] = 0.f;
    }
    if (q.rssi_part) {
        const unsigned k = (unsigned)tid;
        float a = 0.f;
        for (unsigned f = 0; f < nnew; f++) { const cf cu = A[(f + 1) * M + k]; a += hypotf(cu.x, cu.y); }
        q.rssi_part[(size_t)wg * M + k] = a;
    }
}

template <int M, int FPW>
static int launch_fft_disc(hipStream_t st, const pmr_chan_params *p, const cf *Xg, unsigned *ntiles_out)
{
    const unsigned ntiles = (p->ns + FPW - 2) / (FPW - 1);
    if (ntiles_out) *ntiles_out = ntiles;
    const size_t lds = ((size_t)FPW * M + M / 2) * sizeof(cf);
    static pmr_attr_flags attr_set{0};
    if (lds > 64 * 1024 && pmr_attr_needed(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fft_disc<M, FPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    PMR_KLAUNCH((k_fft_disc<M, FPW>), dim3(ntiles), dim3(256), lds, st, *p, Xg);
    return (int)hipGetLastError();
}

extern "C" int pmr_channelize_wide_supported(unsigned M, unsigned p, unsigned nco_period)
{
    /* the filter-bank kernel keeps two NCO factors per thread: the table's period must divide 2 M (it is exactly 2 M) */
    return p == PW_P && (M == 64 || M == 256 || M == 1024 || M == 4096) && nco_period && (2 * M) % nco_period == 0;
}

/* scratch: (ns_max + 1) * M complex floats */
extern "C" int pmr_launch_channelize_wide(pmr_stream_t s, const pmr_chan_params *p, void *x_scratch, unsigned *ntiles_out)
{
    if (ntiles_out) *ntiles_out = 0;
    if (!p->ns) return 0;
    hipStream_t st = (hipStream_t)s;
    if (p->M == 256) {
        const bool small = p->ns < PF_SMALL_NS;
        const unsigned G = small ? PF_G_SMALL : PF_G;
        const unsigned ntiles = (p->ns + G - 1) / G;
        if (ntiles_out) *ntiles_out = ntiles;
        const bool fix = p->fix.V != nullptr;
        const size_t lds = ((size_t)(G + 1) * 256 + 128) * sizeof(cf) + (fix ? pmr_carry_lds_floats(p->fix) * sizeof(float) : 0);
        if (small) {
            if (fix) PMR_KLAUNCH((k_channelize_fused256<true, PF_G_SMALL, PF_RB_SMALL>), dim3(ntiles), dim3(256), lds, st, *p);
            else PMR_KLAUNCH((k_channelize_fused256<false, PF_G_SMALL, PF_RB_SMALL>), dim3(ntiles), dim3(256), lds, st, *p);
        } else {
            if (lds > 64 * 1024) {                              /* (more than 30 frames per workgroup) */
                static pmr_attr_flags attr_set{0};
                if (pmr_attr_needed(attr_set)) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_channelize_fused256<true, PF_G, PF_RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_channelize_fused256<false, PF_G, PF_RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                }
            }
            if (fix) PMR_KLAUNCH((k_channelize_fused256<true, PF_G, PF_RB>), dim3(ntiles), dim3(256), lds, st, *p);
            else PMR_KLAUNCH((k_channelize_fused256<false, PF_G, PF_RB>), dim3(ntiles), dim3(256), lds, st, *p);
        }
        return (int)hipGetLastError();
    }
    if (p->fix.V) return (int)hipErrorInvalidValue;            /* the two-kernel form expects corrected samples */
    unsigned log2M = 0;
    while ((1u << log2M) < p->M) log2M++;
    const unsigned groups = (p->ns + 1 + PW_F - 1) / PW_F;
    const size_t threads = (size_t)groups * p->M;
    PMR_KLAUNCH(k_pfb_wide, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *p, log2M, (cf *)x_scratch);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    switch (p->M) {
    case 64:   return launch_fft_disc<64, 32>(st, p, (const cf *)x_scratch, ntiles_out);
    case 1024: return launch_fft_disc<1024, PW_FPW1024>(st, p, (const cf *)x_scratch, ntiles_out);   /* (three rows per workgroup -- 25 % fewer
                  transforms, half the workgroups -- measured neutral at cfg5, round 3) */
    case 4096: return launch_fft_disc<4096, 2>(st, p, (const cf *)x_scratch, ntiles_out);
    }
    return (int)hipErrorInvalidValue;
}

/* ---- which channelizers subtract the front end's dc carry at load (pmr_carry_fix) ---- */
extern "C" int pmr_channelize_carry_at_load(unsigned M, unsigned p, unsigned nco_period, int chan_small, int chan_wide,
                                            unsigned adv_q, unsigned TQ)
{
    /* k_channelize_win<16, 26, true>: a thread's staged samples are 256 outputs = 16 frames apart, NOV = 2 */
    if (chan_small) return M == 16 && p == 26 && nco_period && 32u % nco_period == 0 && 16u * adv_q < 2u * TQ;
    if (chan_wide) return M == 256 && p == PW_P && adv_q < 2 * TQ;                     /* k_channelize_fused256<true>: NOV = 2 */
    return 0;
}

extern "C" unsigned pmr_channelize_carry_nv(unsigned M, unsigned adv_q, unsigned TQ)
{
    /* frames a workgroup stages: M = 16: up to 256 + 25 rows (k_channelize_win: 240 + 25); M = 256: 9 + 25 rows */
    const unsigned rows = M == 16 ? 256u + 26u : (unsigned)((PF_G > PF_G_SMALL ? PF_G : PF_G_SMALL) + 1 + PW_P);
    return (rows * adv_q) / TQ + 3u;
}
