// pmr_channelize_small.hip -- channelizer + discriminator for SMALL M (M = 16, the PMR446 case) on gfx950.
//
// reference: NCO shift src/sdr_pmr446.c:808-812, firpfbch_crcf_analyzer_execute :814, transpose :819-821,
// freqdem :881.  Sums: tests/chain_model.py  (X_c[t] = sum_k taps_t[k][c] * xm[(t+k)*M + c], y = FFT(X)).
//
// Mapping ("commutator and its small M-point FFT fused per thread"): one thread owns TWO consecutive frames
// with all M polyphase branches in registers (2*M complex accumulators), so
//   * the NCO-mixed samples are staged once per workgroup in LDS (16-byte coalesced HBM loads, 144-byte
//     padded frame rows => conflict-free ds_read_b128),
//   * every tap is a wave-uniform scalar (s_load), each LDS sample feeds two frames,
//   * the M-point FFT runs in registers (radix-2 DIT, same butterfly order and twiddles as the oracle),
//   * the discriminator needs the previous frame: frame B uses frame A of the same thread, frame A uses
//     the neighbour thread's frame B through one LDS exchange.  Local frame 0 of a tile is the frame before
//     its range, recomputed so tiles are independent.
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

static constexpr unsigned brev_c(unsigned v, int bits)
{
    unsigned r = 0;
    for (int b = 0; b < bits; b++) if (v & (1u << b)) r |= 1u << (bits - 1 - b);
    return r;
}

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

#ifndef CS_NT
#define CS_NT 256
#endif
#define CS_FPT 2                          /* frames per thread */

template <int M>
__global__ __launch_bounds__(CS_NT) void k_channelize_small(pmr_chan_params q)
{
    constexpr int L2M = log2c<M>::v;
    constexpr int FS = M + 2;                             // padded frame row in LDS (cf elements): M*8 + 16 bytes
    constexpr int NFT = CS_NT * CS_FPT;                   // frames computed per tile (local frame 0 = frame t0-1)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf *xs = reinterpret_cast<cf *>(smem);                // [(NFT + p - 1)][FS]
    const cf *__restrict__ xr = (const cf *)q.xr;
    const cf *__restrict__ nco_cs = (const cf *)q.nco_cs;
    const cf *__restrict__ fft_tw = (const cf *)q.fft_tw;
    const float *__restrict__ taps_t = q.taps_t;
    const unsigned p = q.p, ns = q.ns, nco_mask = q.nco_period - 1;
    const float fm_ref = q.fm_ref;
    cf *__restrict__ chan_out = (cf *)q.chan_out;
    const unsigned chan_stride = q.chan_stride;
    float *__restrict__ rssi_part = q.rssi_part;

    const int tid = threadIdx.x;
    const long t0 = (long)blockIdx.x * (NFT - 1);         // first NEW frame of this tile, relative to q.frame0
    // local frame l <-> absolute frame frame0 + t0 - 1 + l; it needs absolute frames (.. - (p-1)) .. itself
    const unsigned nfl = NFT + p - 1;                     // frames staged
    const long long s_base = ((long long)q.frame0 + t0 - (long long)p) * M;   // absolute index of the first staged sample

    // ---- stage: HBM ring -> NCO mix -> LDS (two samples per lane per load) ----
    {
        const unsigned units = nfl * (M / 2);
        // NCO phase index of a lane's sample pair is the same for every iteration: 2*CS_NT is a multiple of the period
        const unsigned i0 = ((unsigned)s_base + 2u * tid) & nco_mask;
        const cf c0 = nco_cs[i0], c1 = nco_cs[(i0 + 1) & nco_mask];
        const long long xr_end = (long long)q.xr_end;
        // Batches of CS_SB units per thread: all ring loads of a batch are issued before anything consumes them -- a plain
        // one-unit-per-iteration loop pays the full L2 latency of its load 17 times in a row.
        constexpr int CS_SB = 6;
        for (unsigned u0 = tid; u0 < units; u0 += CS_SB * CS_NT) {
            float4 v[CS_SB];
#pragma unroll
            for (int k = 0; k < CS_SB; k++) {
                const unsigned u = u0 + k * CS_NT;
                const long long a = s_base + 2 * (long long)u;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u < units) {
                    if (a >= 0 && a + 1 < xr_end) v[k] = *reinterpret_cast<const float4 *>(xr + ((unsigned long long)a & q.xr_mask));
                    else if (a >= 0 && a < xr_end) { const cf s1 = xr[(unsigned long long)a & q.xr_mask]; v[k].x = s1.x; v[k].y = s1.y; }
                }
            }
#pragma unroll
            for (int k = 0; k < CS_SB; k++) {
                const unsigned u = u0 + k * CS_NT;
                if (u < units) {
                    const float4 w = v[k];
                    float4 o;
                    o.x = fmaf(w.x, c0.x, w.y * c0.y);        // x * conj(e^{j theta})
                    o.y = fmaf(w.y, c0.x, -(w.x * c0.y));
                    o.z = fmaf(w.z, c1.x, w.w * c1.y);
                    o.w = fmaf(w.w, c1.x, -(w.z * c1.y));
                    const unsigned f = (2 * u) >> L2M, c = (2 * u) & (M - 1);
                    *reinterpret_cast<float4 *>(xs + f * FS + c) = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- polyphase filter bank: two frames per thread, taps wave-uniform ----
    cf XA[M], XB[M];
#pragma unroll
    for (int c = 0; c < M; c++) { XA[c] = cfm(0.f, 0.f); XB[c] = cfm(0.f, 0.f); }
    {
        const cf *row = xs + (size_t)(CS_FPT * tid) * FS;
        for (unsigned j = 0; j <= p; j++) {   // buffer frame (local 2*tid + j) feeds A with tap j, B with tap j-1
            cf s[M];
#pragma unroll
            for (int c = 0; c < M; c += 2) {
                const float4 v = *reinterpret_cast<const float4 *>(row + j * FS + c);
                s[c] = cfm(v.x, v.y); s[c + 1] = cfm(v.z, v.w);
            }
            if (j < p) {
                const float *ta = taps_t + j * M;
#pragma unroll
                for (int c = 0; c < M; c++) XA[c] = cfma(ta[c], s[c], XA[c]);
            }
            if (j >= 1) {
                const float *tb = taps_t + (j - 1) * M;
#pragma unroll
                for (int c = 0; c < M; c++) XB[c] = cfma(tb[c], s[c], XB[c]);
            }
        }
    }

    // ---- M-point FFT in registers (bit-reversed load order is a compile-time renaming) ----
    cf YA[M], YB[M];
#pragma unroll
    for (int c = 0; c < M; c++) { YA[brev_c(c, L2M)] = XA[c]; YB[brev_c(c, L2M)] = XB[c]; }
    fft_dit<M>(YA, fft_tw);
    fft_dit<M>(YB, fft_tw);

    // ---- previous frame for frame A: neighbour thread's frame B, through LDS ----
    __syncthreads();                                      // everyone is done reading the staged samples
    {
        cf *ex = xs + (size_t)tid * FS;
#pragma unroll
        for (int c = 0; c < M; c += 2)
            *reinterpret_cast<float4 *>(ex + c) = make_float4(YB[c].x, YB[c].y, YB[c + 1].x, YB[c + 1].y);
    }
    __syncthreads();
    cf PV[M];
    if (tid > 0) {
        const cf *ex = xs + (size_t)(tid - 1) * FS;
#pragma unroll
        for (int c = 0; c < M; c += 2) {
            const float4 v = *reinterpret_cast<const float4 *>(ex + c);
            PV[c] = cfm(v.x, v.y); PV[c + 1] = cfm(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int c = 0; c < M; c++) PV[c] = cfm(0.f, 0.f);   // unused: local frame 0 produces no output
    }

    // ---- discriminator (:881) + tap-offs ----
    const long tA = t0 - 1 + CS_FPT * tid, tB = tA + 1;   // global frame numbers of this thread
    const bool outA = tid > 0 && tA < (long)ns;
    const bool outB = tB < (long)ns;
    if (outA) {
        float *o = q.fm + ((unsigned long long)(q.frame0 + tA) & q.fm_mask) * M;
#pragma unroll
        for (int k = 0; k < M; k += 4) {
            float4 r;
            float *rr = &r.x;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const cf pv = PV[k + i], cu = YA[k + i];
                rr[i] = pmr_arg(fmaf(pv.x, cu.y, -(pv.y * cu.x)), fmaf(pv.x, cu.x, pv.y * cu.y)) * fm_ref;
                if (tA == 0 && q.reset_flags && q.reset_flags[k + i]) rr[i] = 0.f;    // freqdem_reset: arg(0) = 0
            }
            *reinterpret_cast<float4 *>(o + k) = r;
        }
    }
    if (outB) {
        float *o = q.fm + ((unsigned long long)(q.frame0 + tB) & q.fm_mask) * M;
#pragma unroll
        for (int k = 0; k < M; k += 4) {
            float4 r;
            float *rr = &r.x;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const cf pv = YA[k + i], cu = YB[k + i];
                rr[i] = pmr_arg(fmaf(pv.x, cu.y, -(pv.y * cu.x)), fmaf(pv.x, cu.x, pv.y * cu.y)) * fm_ref;
                if (tB == 0 && q.reset_flags && q.reset_flags[k + i]) rr[i] = 0.f;
            }
            *reinterpret_cast<float4 *>(o + k) = r;
        }
    }
    if (chan_out) {
#pragma unroll
        for (int k = 0; k < M; k++) {
            if (outA) chan_out[(size_t)k * chan_stride + tA] = YA[k];
            if (outB) chan_out[(size_t)k * chan_stride + tB] = YB[k];
        }
    }
    if (rssi_part) {
        // per-channel sum of |y| over the tile's new frames: wave butterfly, then one LDS pass
        __syncthreads();
        float *red = reinterpret_cast<float *>(xs);       // [CS_NT/64][M]
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int k = 0; k < M; k++) {
            float a = (outA ? hypotf(YA[k].x, YA[k].y) : 0.f) + (outB ? hypotf(YB[k].x, YB[k].y) : 0.f);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
            if (lane == 0) red[wave * M + k] = a;
        }
        __syncthreads();
        if (tid < M) {
            float a = 0.f;
            for (int w = 0; w < CS_NT / 64; w++) a += red[w * M + tid];
            rssi_part[(size_t)blockIdx.x * M + tid] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Second mapping for M = 16 (the default): staged window + sliding-window filter bank + per-frame register FFT.
//   pass 0  the tile's NFT + 25 input rows go from the resampled ring to LDS ONCE: 8-byte coalesced loads (a wave = four 128-byte
//           rows), the front end's dc carry subtracted (FIX) and the NCO factor applied on the way -- 17.6 samples per thread.
//           (Round 3's first form read the rows straight from the ring in pass 1: every sample was loaded, corrected and mixed by
//           each of the 2.6 threads that use it -- the carry arithmetic alone was 37 % of the kernel's vector instructions.)
//   pass 1  thread = (channel c, group of F consecutive frames): the 26 branch taps sit in registers, the group's F + 25 rows
//           come from LDS (row pitch 17: the four groups of a wave meet both bank halves), every sample feeds up to F frames.
//           The bank outputs X[frame][c] then OVERWRITE the staged rows (barrier in between): 36 KB + the carry tables, four
//           tiles per CU -- which is why a tile is 15 x 16 = 240 frames, not 256.
//   pass 2  thread = frame: FFT-16 in registers, previous frame through LDS, discriminator, stores.
// Same products and the same oldest-first accumulation order as k_channelize_small / the oracle.
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
    const long tA = t0 - 1 + tid;                         // frame of this thread, relative to q.frame0
    const bool outA = tid > 0 && mine && tA < (long)ns;
    if (outA) {
        const cf *pvrow = Xs + (size_t)(tid - 1) * FS;
        float *o = q.fm + ((unsigned long long)(q.frame0 + tA) & q.fm_mask) * M;
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

static unsigned pmr_channelize_small_tiles(unsigned ns) { return (ns + CS_NT * CS_FPT - 2) / (CS_NT * CS_FPT - 1); }   /* pair kernel */

extern "C" int pmr_channelize_small_supported(unsigned M, unsigned p, unsigned nco_period)
{
    return M == 16 && p >= 2 && p <= 64 && nco_period && (2u * CS_NT) % nco_period == 0;
}

extern "C" int pmr_launch_channelize_small(pmr_stream_t s, const pmr_chan_params *p, unsigned *ntiles_out, int pair)
{
    /* pair = two frames per thread (PMR_CHANNELIZER_SMALL=pair); the sliding-window kernel keeps two NCO factors per thread */
    const bool win = !pair && p->p == 26 && (2u * p->M) % p->nco_period == 0;
    /* few frames: 4 instead of 16 frames per (channel, group) item -- more, shorter workgroups (latency of the synchronous calls) */
    const bool fine = win && p->ns < 16u * CW_NT;
    const unsigned nft = fine ? (unsigned)cw_geom<CW_F_SMALL>::NFT : (unsigned)cw_geom<CW_F>::NFT;
    const unsigned ntiles = win ? (p->ns + nft - 2) / (nft - 1) : pmr_channelize_small_tiles(p->ns);
    if (ntiles_out) *ntiles_out = ntiles;
    if (!p->ns) return 0;
    if (p->M != 16) return (int)hipErrorInvalidValue;
    if (win) {
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
    if (p->fix.V) return (int)hipErrorInvalidValue;            /* the two-frames-per-thread kernel expects corrected samples */
    const size_t lds = (size_t)(CS_NT * CS_FPT + p->p - 1) * (p->M + 2) * sizeof(cf);
    static pmr_attr_flags attr_set{0};
    if (pmr_attr_needed(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize_small<16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    PMR_KLAUNCH(k_channelize_small<16>, dim3(ntiles), dim3(CS_NT), lds, (hipStream_t)s, *p);
    return (int)hipGetLastError();
}
