// pmr_channelize_small.hip -- channelizer + discriminator for SMALL M (M = 16, the PMR446 case) on gfx950.
//
// reference: NCO shift src/sdr_pmr446.c:808-812, firpfbch_crcf_analyzer_execute :814, transpose :819-821,
// freqdem :881.  Sums: tests/chain_model.py  (X_c[t] = sum_k taps_t[k][c] * xm[(t+k)*M + c], y = FFT(X)).
//
// One kernel, k_channelize_win (mapping below): the 16-channel bank of the PMR446 plan (26 branch taps).  Any other small
// configuration (M = 16 with another prototype length, M = 4, 8, 32 ...) takes the generic k_channelize (pmr_kernels.hip).
// (Round 1's two-frames-per-thread mapping lived here until round 4; it lost to the sliding-window form in round 3.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pmr_kernels.h"
#include "pmr_carry_load.hpp"

// complex = ext-vector pair: real-tap MACs and butterflies map onto v_pk_fma_f32 / v_pk_add_f32 (~1.8x the FLOP rate of
// the scalar forms on gfx950, tools/ubench/valu_rate.hip)
typedef float cf __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ cf cfm(float r, float i) { return cf{r, i}; }
static __device__ __forceinline__ cf cfma(float h, cf x, cf acc) { return __builtin_elementwise_fma(cf{h, h}, x, acc); }

template <int M> struct log2c { static constexpr int v = 1 + log2c<M / 2>::v; };
template <> struct log2c<1> { static constexpr int v = 0; };

static __device__ __forceinline__ unsigned brev_rt(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// in-register forward DFT, radix-2 decimation in time; x must already be in bit-reversed order
template <int M>
static __device__ __forceinline__ void fft_dit(cf (&x)[M], const cf *__restrict__ tw /*[M/2], wave-uniform*/)
{
#pragma unroll
    for (int len = 2; len <= M; len <<= 1) {
        const int half = len >> 1, tstep = M / len;
#pragma unroll
        for (int base = 0; base < M; base += len) {
#pragma unroll
            for (int k = 0; k < half; k++) {
                const cf a = x[base + k], b = x[base + k + half];
                cf t;
                if (k == 0) {
                    t = b;                                               // w = 1 (known at compile time: loops are unrolled)
                } else if (4 * k == len) {
                    t = cf{b.y, -b.x};                                   // w = -j
                } else {
                    const cf w = tw[k * tstep];                          // (cos, sin) of -2 pi k / len, wave-uniform
                    t = __builtin_elementwise_fma(cf{b.x, b.x}, w, cf{b.y, b.y} * cf{-w.y, w.x});
                }
                x[base + k] = a + t;
                x[base + k + half] = a - t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// M = 16, 26 branch taps: staged window + sliding-window filter bank + per-frame register FFT.
//   pass 0  the tile's NFT + 25 input rows go from the resampled ring to LDS ONCE: 8-byte coalesced loads (a wave = four 128-byte
//           rows), the front end's dc carry subtracted (FIX) and the NCO factor applied on the way -- 17.6 samples per thread.
//           (Round 3's first form read the rows straight from the ring in pass 1: every sample was loaded, corrected and mixed by
//           each of the 2.6 threads that use it -- the carry arithmetic alone was 37 % of the kernel's vector instructions.)
//   pass 1  thread = (channel c, group of F consecutive frames): the 26 branch taps sit in registers, the group's F + 25 rows
//           come from LDS (row pitch 17: the four groups of a wave meet both bank halves), every sample feeds up to F frames.
//           The bank outputs X[frame][c] then OVERWRITE the staged rows (barrier in between): 36 KB + the carry tables, four
//           tiles per CU -- which is why a tile is 15 x 16 = 240 frames, not 256.
//   pass 2  thread = frame: FFT-16 in registers, previous frame through LDS, discriminator, stores.
// Same products and the same oldest-first accumulation order as the generic k_channelize / the oracle.
// ---------------------------------------------------------------------------------------------------------------
#ifndef CW_NT
#define CW_NT 256                         /* threads per tile */
#endif
#define CW_F 16                          /* frames per (channel, group) work item: big blocks */
#define CW_F_SMALL 4                     /* ... blocks of a few thousand frames (the reference's 100 000-sample blocks: 1220 frames): 63-frame
                                            tiles, 20 workgroups instead of 5, a third of the serial work per thread */
template <int F> struct cw_geom {
    static constexpr int NG = F == CW_F ? CW_NT / 16 - 1 : CW_NT / 16;       // (channel, group) groups per tile
    static constexpr int NFT = NG * F;                                       // frames per tile (local frame 0 = frame t0 - 1, recomputed)
    static constexpr int NR = NFT + 26 - 1;                                  // staged rows
    static constexpr int SS = 16 + 1, FS = 16 + 2;                           // pitches (cf) of the staged rows / the X rows
    static constexpr int LDS_CF = NR * SS > NFT * FS ? NR * SS : NFT * FS;
};

// FIX: the front end's dc carry is subtracted from the samples as they are staged (pmr_carry_fix / pmr_carry_load.hpp): a thread's
// samples are CW_NT = 256 outputs apart, i.e. 16 frames ~ 400 decimated samples at cfg2 -- less than two front-end tiles (NOV = 2).
// (Letting the workgroup that finishes last also reduce the tiles' RSSI partial sums -- a launch less for the small synchronous
// calls -- was measured: the device-scope fences it needs cost 6 us, the separate k_rssi_finish launch 4.6.)
template <int M, int P, bool FIX, int F>
__global__ __launch_bounds__(CW_NT, (CW_NT <= 256 ? 4 : CW_NT <= 512 ? 2 : 1)) void k_channelize_win(pmr_chan_params q)
{
    static_assert(M == 16 && P == 26, "geometry (cw_geom) is written for the PMR446 bank");
    constexpr int L2M = log2c<M>::v;
    typedef cw_geom<F> G;
    constexpr int FS = G::FS, SS = G::SS, NFT = G::NFT, NR = G::NR, NG = G::NG;
    static_assert(CW_NT % M == 0 && NFT <= CW_NT && (CW_NT % (2 * M)) == 0, "pass 2 gives every frame a thread; one NCO factor per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *Ss = reinterpret_cast<cf *>(smem);                // [NR][SS] staged, corrected, mixed samples
    cf *Xs = Ss;                                          // [NFT][FS] bank outputs, over them
    const cf *__restrict__ xr = (const cf *)q.xr;
    const cf *__restrict__ nco_cs = (const cf *)q.nco_cs;
    const cf *__restrict__ fft_tw = (const cf *)q.fft_tw;
    const unsigned ns = q.ns, nco_mask = q.nco_period - 1;
    const float fm_ref = q.fm_ref;
    const int tid = threadIdx.x;
    const unsigned wg = pmr_xcd_contiguous(blockIdx.x, gridDim.x);
    const long t0 = (long)wg * (NFT - 1);                 // first NEW frame of this tile, relative to q.frame0
    const long long fb = (long long)q.frame0 + t0 - (long long)P;      // absolute frame of staged row 0
    pmr_carry_lds ct;
    if constexpr (FIX) {
        ct = pmr_carry_setup<CW_NT>(q.fix, reinterpret_cast<float *>(Ss + G::LDS_CF), fb * M - (long long)q.fix.pos0, tid);
        __syncthreads();
    }

    // ---- pass 0: ring -> (carry fix) -> NCO mix -> LDS, sample i = tid + CW_NT * it of the tile's NR * M ----
    {
        constexpr int NIT = (NR * M + CW_NT - 1) / CW_NT, CB = 6;
        // low 32 bits of the absolute sample index are all the ring / NCO masks need; indices before the stream start wrap into the
        // zero-initialised top of the ring.  CW_NT is a multiple of the NCO table's period (launcher): one factor per thread
        const unsigned a0 = (unsigned)((unsigned long long)fb * (unsigned long long)M) + (unsigned)tid, xr_mask32 = (unsigned)q.xr_mask;
        const cf cs = nco_cs[a0 & nco_mask];
        pmr_carry_state cst;
        const unsigned long long dph = (unsigned long long)CW_NT * q.fix.step;
        if constexpr (FIX) cst = pmr_carry_init(q.fix, ct, fb * (long long)M + (long long)tid - (long long)q.fix.pos0);
#pragma unroll
        for (int i0 = 0; i0 < NIT; i0 += CB) {
            cf v[CB];
#pragma unroll
            for (int u = 0; u < CB; u++) {
                const int it = i0 + u;
                if (it < NIT) {
                    const float2 w = reinterpret_cast<const float2 *>(xr)[(a0 + (unsigned)(CW_NT * it)) & xr_mask32];
                    v[u] = cfm(w.x, w.y);
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // the chunk's loads stay together, ahead of everything that consumes them
#pragma unroll
            for (int u = 0; u < CB; u++) {
                const int it = i0 + u;
                if (it < NIT) {
                    cf x = v[u];
                    if constexpr (FIX) x = pmr_carry_apply<2>(q.fix, ct, cst, x, CW_NT, dph);
                    x = cfm(fmaf(x.x, cs.x, x.y * cs.y), fmaf(x.y, cs.x, -(x.x * cs.y)));       // x * conj(e^{j theta})
                    const int i = tid + CW_NT * it;
                    if (CW_NT * (it + 1) <= NR * M || i < NR * M) Ss[(i >> L2M) * SS + (i & (M - 1))] = x;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
#if defined(CW_STOP) && CW_STOP == 0      /* timing experiments (tools/ab_libs.py): the kernel up to a pass boundary.  WRONG results */
    if (q.ns != 0xffffffffu) return;
#endif

    // ---- pass 1: polyphase bank, X[f][brev(c)] ----
    {
        const unsigned c = tid & (M - 1), g = tid >> L2M, f0 = g * F;
        const bool work = NG * M == CW_NT || g < (unsigned)NG;
        cf acc[F];
#pragma unroll
        for (int f = 0; f < F; f++) acc[f] = cfm(0.f, 0.f);
        if (work) {
            float h[P];
#pragma unroll
            for (int k = 0; k < P; k++) h[k] = q.taps_t[k * M + c];
            const cf *row = Ss + (size_t)f0 * SS + c;     // staged row f0 + r <-> frame t0 - 1 - (P - 1) + f0 + r  (k = 0 oldest)
#pragma unroll
            for (int r = 0; r < F + P - 1; r++) {
                const cf x = row[r * SS];
#pragma unroll
                for (int f = (r - P + 1 > 0 ? r - P + 1 : 0); f <= (r < F - 1 ? r : F - 1); f++)
                    acc[f] = cfma(h[r - f], x, acc[f]);
            }
        }
        __syncthreads();                                  // every staged row has been read: the X rows go over them
        if (work) {
            const unsigned rc = brev_rt(c, L2M);
#pragma unroll
            for (int f = 0; f < F; f++) Xs[(f0 + f) * FS + rc] = acc[f];
        }
    }
    __syncthreads();

#if defined(CW_STOP) && CW_STOP == 1
    if (q.ns != 0xffffffffu) return;
#endif
    // ---- pass 2: thread = local frame tid; FFT in registers ----
    const bool mine = NFT == CW_NT || tid < NFT;          // the first NFT threads own a frame
    cf Y[M];
    {
        const cf *row = Xs + (size_t)(mine ? tid : 0) * FS;
#pragma unroll
        for (int c = 0; c < M; c += 2) {
            const float4 v = *reinterpret_cast<const float4 *>(row + c);
            Y[c] = cfm(v.x, v.y); Y[c + 1] = cfm(v.z, v.w);
        }
    }
    fft_dit<M>(Y, fft_tw);
    __syncthreads();                                      // every X row has been read
    if (mine) {
        cf *ex = Xs + (size_t)tid * FS;
#pragma unroll
        for (int c = 0; c < M; c += 2)
            *reinterpret_cast<float4 *>(ex + c) = make_float4(Y[c].x, Y[c].y, Y[c + 1].x, Y[c + 1].y);
    }
    __syncthreads();
#if defined(CW_STOP) && CW_STOP == 2
    if (q.ns != 0xffffffffu) return;
#endif
    const long tA = t0 - 1 + tid;                         // frame of this thread, relative to q.frame0
    const bool outA = tid > 0 && mine && tA < (long)ns;
    if (outA) {
        const cf *pvrow = Xs + (size_t)(tid - 1) * FS;
        float *o = q.fm + ((unsigned long long)(q.frame0 + tA) & q.fm_mask & PMR_EXP_ROW_AND) * M;
#pragma unroll
        for (int k = 0; k < M; k += 4) {
            const float4 p0 = *reinterpret_cast<const float4 *>(pvrow + k), p1 = *reinterpret_cast<const float4 *>(pvrow + k + 2);
            const cf pv[4] = {cfm(p0.x, p0.y), cfm(p0.z, p0.w), cfm(p1.x, p1.y), cfm(p1.z, p1.w)};
            float4 r;
            float *rr = &r.x;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const cf cu = Y[k + i];
                rr[i] = pmr_arg(fmaf(pv[i].x, cu.y, -(pv[i].y * cu.x)), fmaf(pv[i].x, cu.x, pv[i].y * cu.y)) * fm_ref;
            }
            *reinterpret_cast<float4 *>(o + k) = r;
        }
        if (tA == 0 && q.reset_flags) {                  // freqdem_reset: arg(0) = 0 -- one thread of the call, after its stores (a test
#pragma unroll                                           //  per element sat in all sixteen arg() chains as a taken branch)
            for (int k = 0; k < M; k++) if (q.reset_flags[k]) o[k] = 0.f;
        }
        cf *__restrict__ chan_out = (cf *)q.chan_out;
        if (chan_out) {
#pragma unroll
            for (int k = 0; k < M; k++) chan_out[(size_t)k * q.chan_stride + tA] = Y[k];
        }
    }
    if (q.rssi_part) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(Xs);       // [CW_NT/64][M]
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int k = 0; k < M; k++) {
            float a = outA ? hypotf(Y[k].x, Y[k].y) : 0.f;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
            if (lane == 0) red[wave * M + k] = a;
        }
        __syncthreads();
        if (tid < M) {
            float a = 0.f;
            for (int w = 0; w < CW_NT / 64; w++) a += red[w * M + tid];
            q.rssi_part[(size_t)wg * M + tid] = a;
        }
    }
}

extern "C" int pmr_channelize_small_supported(unsigned M, unsigned p, unsigned nco_period)
{
    /* the sliding-window kernel keeps one NCO factor per thread: the table's period must divide the tile's thread count and 2 M */
    return M == 16 && p == 26 && nco_period && CW_NT % nco_period == 0 && (2u * M) % nco_period == 0;
}

extern "C" int pmr_launch_channelize_small(pmr_stream_t s, const pmr_chan_params *p, unsigned *ntiles_out)
{
    /* few frames: 4 instead of 16 frames per (channel, group) item -- more, shorter workgroups (latency of the synchronous calls) */
    const bool fine = p->ns < 16u * CW_NT;
    const unsigned nft = fine ? (unsigned)cw_geom<CW_F_SMALL>::NFT : (unsigned)cw_geom<CW_F>::NFT;
    const unsigned ntiles = (p->ns + nft - 2) / (nft - 1);
    if (ntiles_out) *ntiles_out = ntiles;
    if (!p->ns) return 0;
    if (!pmr_channelize_small_supported(p->M, p->p, p->nco_period)) return (int)hipErrorInvalidValue;
    const bool fix = p->fix.V != nullptr;
    const size_t lds_w = (size_t)(fine ? cw_geom<CW_F_SMALL>::LDS_CF : cw_geom<CW_F>::LDS_CF) * sizeof(cf) +
                         (fix ? pmr_carry_lds_floats(p->fix) * sizeof(float) : 0);
    hipStream_t st = (hipStream_t)s;
    static pmr_attr_flags attr_w{0};
    if (lds_w > 64 * 1024 && pmr_attr_needed(attr_w)) {      /* only -DCW_NT=1024 experiment builds get here */
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_win<16, 26, false, CW_F>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_win<16, 26, true, CW_F>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (fine) {
        if (fix) PMR_KLAUNCH((k_channelize_win<16, 26, true, CW_F_SMALL>), dim3(ntiles), dim3(CW_NT), lds_w, st, *p);
        else PMR_KLAUNCH((k_channelize_win<16, 26, false, CW_F_SMALL>), dim3(ntiles), dim3(CW_NT), lds_w, st, *p);
    } else {
        if (fix) PMR_KLAUNCH((k_channelize_win<16, 26, true, CW_F>), dim3(ntiles), dim3(CW_NT), lds_w, st, *p);
        else PMR_KLAUNCH((k_channelize_win<16, 26, false, CW_F>), dim3(ntiles), dim3(CW_NT), lds_w, st, *p);
    }
    return (int)hipGetLastError();
}
