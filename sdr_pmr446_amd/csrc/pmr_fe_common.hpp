// pmr_fe_common.hpp -- device helpers shared by the front-end kernels (pmr_frontend.hip: run-time-parameterised tile
// kernel + carry kernels; pmr_fe_fast.hip: the specialised kernels for the cascades the As = 60 dB design produces).
// Reference stage: dc-block + msresamp_crcf, src/sdr_pmr446.c:795-796.
#ifndef PMR_FE_COMMON_HPP
#define PMR_FE_COMMON_HPP

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "pmr_kernels.h"

// complex sample = clang ext-vector pair: (re, im) arithmetic with a real scalar tap maps onto v_pk_fma_f32 with the tap
// broadcast from one SGPR.  Measured on MI355X (tools/ubench/valu_rate.hip): v_fma_f32 peaks at ~67 TFLOP/s,
// v_pk_fma_f32 at ~115-120 TFLOP/s, so packed math is worth ~1.8x wherever the kernel is VALU-bound.
typedef float cf __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ cf cfm(float r, float i) { return cf{r, i}; }
static __device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
static __device__ __forceinline__ cf cadd_scale(cf a, cf b, float s) { return (a + b) * cf{s, s}; }
static __device__ __forceinline__ cf cfma(float h, cf x, cf acc) { return __builtin_elementwise_fma(cf{h, h}, x, acc); }

// Cross-lane moves on the VALU (DPP), not through the LDS crossbar: the kernel is LDS-instruction-bound.
//   row_shr:n (0x110+n) shift inside a 16-lane row; wave_shr:1 (0x138) shift across the whole wave;
//   row_bcast:15 / row_bcast:31 (0x142 / 0x143) lane 15 / 31 of a row to every lane of the next row(s).
// Lanes without a source (or masked rows) receive `old` = 0.
template <int CTRL, int ROW_MASK = 0xF>
static __device__ __forceinline__ float dpp0(float src)
{
    // all rows enabled: bound_ctrl:1 makes the lanes without a source read 0 by themselves -- ONE v_mov_b32_dpp instead of a v_mov
    // that zeroes the destination first plus the DPP move (the dc scan and the half-band halo are 34 such moves per thread and tile)
    if constexpr (ROW_MASK == 0xF)
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), CTRL, 0xF, 0xF, true));
    else
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
static __device__ __forceinline__ cf dpp0c(cf v) { return cf{dpp0<CTRL, ROW_MASK>(v.x), dpp0<CTRL, ROW_MASK>(v.y)}; }

// LDS layout L(G): element e lives at e + e/G (one 8-byte pad per G elements) so that threads whose chunks
// are G elements apart hit distinct banks with ds_read_b64 / ds_write_b64 (stride 2G+2 dwords, gcd with 64 = 2).
template <int G> static __device__ __forceinline__ int lidx(int e) { return e + e / G; }
static __device__ __forceinline__ int lidx_rt(int e, int g_shift) { return e + (e >> g_shift); }

/* experiment hooks (tools/variant_kstats.sh): FE_OUT_AND masks the output ring index (all tiles store into one small window: no
 * HBM write traffic; WRONG results), FE_OUT_NT makes the 16-byte output stores non-temporal */
#ifndef FE_OUT_AND
#define FE_OUT_AND (~0ull)
#endif
typedef float fe_f4 __attribute__((ext_vector_type(4)));
#if defined(FE_OUT_NT)
#define FE_STORE4(ptr, val) do { const float4 v_ = (val); __builtin_nontemporal_store(fe_f4{v_.x, v_.y, v_.z, v_.w}, reinterpret_cast<fe_f4 *>(ptr)); } while (0)
#elif defined(FE_OUT_SC)        /* write-through to system scope: nothing of the ring stays dirty in L2 for the end-of-kernel release */
#define FE_STORE4(ptr, val) do { const float4 v_ = (val); const fe_f4 w_ = {v_.x, v_.y, v_.z, v_.w};                                \
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(ptr), "v"(w_) : "memory"); } while (0)
#elif defined(FE_OUT_SKIP)      /* everything but the store itself (the condition is never true) */
#define FE_STORE4(ptr, val) do { if (p.n_in == 0xfffffffeu) *reinterpret_cast<float4 *>(ptr) = (val); } while (0)
#else
#define FE_STORE4(ptr, val) (*reinterpret_cast<float4 *>(ptr) = (val))
#endif

#define FE_PAD 80      /* zeroed elements in front of the tile buffer: a stage's window left of sample 0, (4 m - 2) (1 + 1/G) elements in layout
                          L(G); largest: m = 12 read in layout L(2) = 69 (the kernels static_assert their own) */

static constexpr int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
template <int G> static constexpr int loff(int e) { return e + fdiv(e, G); }   // layout offset of a constant index

// ceil(num / den) for num < 2^56 and den in [2^24, 2^25] (the resampler step) in INTEGER arithmetic: rinv = floor(2^56 / den)
// clamped to 32 bits (host).  q0 = ((num >> 24) * rinv) >> 32 never exceeds floor(num / den) and is short of it by at most 3
// (truncation of num by < 2^24, of rinv by < 1, of the product by < 1), so the fix-up loop runs <= 3 times.  Every operand is
// wave-uniform where this is used (tile bookkeeping), so the whole thing stays on the scalar unit -- the fp64 division it
// replaces was ~100 vector instructions per thread.
static __device__ __forceinline__ unsigned long long ceil_div_step(unsigned long long num, unsigned den, unsigned rinv)
{
    unsigned long long q = ((num >> 24) * (unsigned long long)rinv) >> 32;
    while ((q + 1) * den <= num) q++;                     // q = floor(num / den)
    return q * den == num ? q : q + 1;
}

// one half-band stage, ping-pong form: reads layout L(2P) from `src`, writes layout L(max(P, 2)) to `dst`, ONE barrier.
//   z1[o] = z0[2o+1-2m] + sum_j h1[j] * z0[2o - 2(2m-1-j)]            (resamp2 decimator)
// Every LDS address is thread base + compile-time constant (floor division keeps that true left of the tile, where the
// reads land in the zero pad / the previous region and only feed outputs inside the halo).
template <int P, int MM>
static __device__ __forceinline__ void hb_stage_pp(const cf *__restrict__ src, cf *__restrict__ dst, int tid,
                                                   int n_threads, const float *h1, float scale)
{
    constexpr int NE = P + 2 * MM - 1, G = 2 * P;
    if (tid < n_threads) {
        const cf *w = src + tid * (G + 1);
        cf we[NE], wd[P];
#pragma unroll
        for (int i = 0; i < NE; i++) we[i] = w[loff<G>(2 * i - (4 * MM - 2))];
#pragma unroll
        for (int p = 0; p < P; p++) wd[p] = w[loff<G>(2 * p + 1 - 2 * MM)];
        cf y[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            cf a = cfm(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 2 * MM; j++) a = cfma(h1[j], we[p + j], a);
            y[p] = cadd_scale(wd[p], a, scale);
        }
        if constexpr (P >= 2) {
            cf *o = dst + tid * (P + 1);               // L(P)
#pragma unroll
            for (int p = 0; p < P; p++) o[p] = y[p];
        } else {
            dst[tid + (tid >> 1)] = y[0];              // L(2)
        }
    }
    __syncthreads();
}

// a stage with ONE output per thread whose result stays in a register (the level-1 kernel's last stage: its outputs go straight to
// the ring, no LDS round trip): hb_stage_pp<1, MM>'s window and accumulation order
template <int MM>
static __device__ __forceinline__ cf hb_stage_out1(const cf *__restrict__ src, int tid, const float *h1, float scale)
{
    constexpr int NE = 2 * MM, G = 2;
    const cf *w = src + tid * (G + 1);
    cf we[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) we[i] = w[loff<G>(2 * i - (4 * MM - 2))];
    const cf wd = w[loff<G>(1 - 2 * MM)];
    cf a = cfm(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 2 * MM; j++) a = cfma(h1[j], we[j], a);
    return cadd_scale(wd, a, scale);
}

// a stage whose INPUT sits in registers (round 6): thread t holds zin[e] = z[PIN t + e] and computes its PIN / 2 outputs
//   y[p] = z[2o + 1 - 2 MM] + sum_j h1[j] z[2o - (4 MM - 2) + 2j],  o = (PIN / 2) t + p     (hb_stage_pp's arithmetic, same order: bit-identical)
// The window reaches LM = ceil((4 MM - 2) / PIN) threads to the left: chained DPP wave_shr:1 moves bring their elements over
// (S[s][e] = element e of thread t - s; the compiler drops the ones no tap uses) instead of a write of every sample to LDS and a
// window of reads back.  Only a wave's first LM lanes need the previous wave's last LM lanes: those PIN LM samples per wave go
// through `xch` ([4 waves][LM][PIN], LDS), ONE barrier.  Wave 0's first lanes read zeros there (left of the tile: halo only).
// Consecutive register stages must use different `xch` areas (a fast wave may publish the next stage's samples while a slow one
// still reads this stage's).
template <int PIN, int MM>
static __device__ __forceinline__ void hb_stage_reg(const cf (&zin)[PIN], cf (&y)[PIN / 2], cf *xch, int lane, int wave,
                                                    const float *h1, float scale)
{
    constexpr int LM = (4 * MM - 2 + PIN - 1) / PIN;
    static_assert(LM <= 8 && PIN >= 2, "hb_stage_reg: the window must stay within a few threads");
    if (lane >= 64 - LM) {
#pragma unroll
        for (int e = 0; e < PIN; e++) xch[(wave * LM + (lane - (64 - LM))) * PIN + e] = zin[e];
    }
    __syncthreads();
    cf S[LM + 1][PIN];
#pragma unroll
    for (int e = 0; e < PIN; e++) S[0][e] = zin[e];
#pragma unroll
    for (int sft = 1; sft <= LM; sft++) {
#pragma unroll
        for (int e = 0; e < PIN; e++) S[sft][e] = dpp0c<0x138>(S[sft - 1][e]);                     // wave_shr:1 (lane 0 <- 0)
    }
    if (wave > 0 && lane < LM) {
#pragma unroll
        for (int sft = 1; sft <= LM; sft++) {
            if (lane < sft) {
                const cf *src = xch + ((wave - 1) * LM + (LM + lane - sft)) * PIN;
#pragma unroll
                for (int e = 0; e < PIN; e++) S[sft][e] = src[e];
            }
        }
    }
#pragma unroll
    for (int pp = 0; pp < PIN / 2; pp++) {
        cf a = cfm(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 2 * MM; j++) {
            const int rel = 2 * pp - (4 * MM - 2) + 2 * j, L = (PIN - 1 - rel) / PIN;              // rel <= PIN - 2: L = ceil(-rel / PIN)
            a = cfma(h1[j], S[L][rel + PIN * L], a);
        }
        const int rd = 2 * pp + 1 - 2 * MM, Ld = (PIN - 1 - rd) / PIN;
        y[pp] = cadd_scale(S[Ld][rd + PIN * Ld], a, scale);
    }
}

// the same stage IN PLACE: read window -> barrier -> write over the input -> barrier.  Two barriers instead of one, half the LDS
// (a level-2 tile then fits beside four level-1 tiles on a CU).
template <int P, int MM>
static __device__ __forceinline__ void hb_stage_ip(cf *__restrict__ buf, int tid, int n_threads, const float *h1, float scale)
{
    constexpr int NE = P + 2 * MM - 1, G = 2 * P;
    cf y[P];
    if (tid < n_threads) {
        const cf *w = buf + tid * (G + 1);
        cf we[NE], wd[P];
#pragma unroll
        for (int i = 0; i < NE; i++) we[i] = w[loff<G>(2 * i - (4 * MM - 2))];
#pragma unroll
        for (int p = 0; p < P; p++) wd[p] = w[loff<G>(2 * p + 1 - 2 * MM)];
#pragma unroll
        for (int p = 0; p < P; p++) {
            cf a = cfm(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 2 * MM; j++) a = cfma(h1[j], we[p + j], a);
            y[p] = cadd_scale(wd[p], a, scale);
        }
    }
    __syncthreads();
    if (tid < n_threads) {
        if constexpr (P >= 2) {
            cf *o = buf + tid * (P + 1);               // L(P)
#pragma unroll
            for (int p = 0; p < P; p++) o[p] = y[p];
        } else {
            buf[tid + (tid >> 1)] = y[0];              // L(2)
        }
    }
    __syncthreads();
}

// Resampler bookkeeping of one tile (wave-uniform): the tile owns decimated samples [qa, qb) and therefore the resampler
// outputs j in [ja, jb) whose input index (phi0 + j*step) >> 24 falls in that range (resamp_crcf phase rule).
struct fe_jrange { unsigned long long ja, jb; };
static __device__ __forceinline__ fe_jrange fe_tile_outputs(const pmr_fe_params &p, unsigned long long qa)
{
    fe_jrange r = {0ull, 0ull};
    unsigned long long qb = qa + (unsigned)p.TQ;
    if (qb > p.Q) qb = p.Q;
    if (qa < qb) {
        const unsigned long long sa = qa << 24, sb = qb << 24;
        r.ja = sa <= p.phi0 ? 0ull : ceil_div_step(sa - p.phi0, p.step, p.step_rinv);
        r.jb = sb <= p.phi0 ? 0ull : ceil_div_step(sb - p.phi0, p.step, p.step_rinv);
        if (r.jb > p.ny) r.jb = p.ny;
        if (r.ja > r.jb) r.ja = r.jb;
    }
    return r;
}

// Phase D of the specialised kernels: arbitrary resampler (firpfb of 256 x 14 taps, closed-form 24-bit phase) out of the last
// stage's output `fin` (layout L(1 << GS)), ring store.  A thread owns the PAIR of adjacent outputs jh + 2 tid, + 1 (jh = ja
// rounded up to an even ring position; the odd head sample, if any, is thread 0's extra job): one 16-byte store per thread
// instead of two 8-byte ones -- only when the tile has more outputs than threads, otherwise one output per thread is the
// shorter phase.  fe_arb_prefetch requests the polyphase taps (two rows of the bank) long before fe_arb_store uses them.
struct fe_arb_plan { unsigned long long ja, jb, jh, j0; bool pairs; int npts; bool head_owned; const float *b0p, *b1p; };
template <int NT, bool HEAD_LAST = false>
static __device__ __forceinline__ fe_arb_plan fe_arb_prepare(const pmr_fe_params &p, unsigned long long qa, int tid)
{
    fe_arb_plan a;
    const fe_jrange r = fe_tile_outputs(p, qa);
    a.ja = r.ja; a.jb = r.jb;
#ifdef FE_NO_PAIRS      /* experiment: one output per thread even in tiles with more outputs than threads (the rest in the slow leftover loop) */
    a.pairs = false;
#else
    a.pairs = a.jb - a.ja > (unsigned long long)NT;
#endif
    a.jh = a.pairs ? a.ja + ((p.out_pos0 + a.ja) & 1ull) : a.ja;
    a.j0 = a.pairs ? a.jh + 2ull * tid : a.ja + tid;
    a.npts = a.pairs ? (a.j0 + 1 < a.jb ? 2 : (a.j0 < a.jb ? 1 : 0)) : (a.j0 < a.jb ? 1 : 0);
    // the odd head sample of a pairs tile (half of them have one) belongs to the LAST thread when that thread has no pair of its own
    // (tiles of up to 2 NT - 2 outputs): its taps are then requested here, long before fe_arb_store -- as thread 0's extra job with
    // taps fetched on the spot it was a dependent global load at the very end of the tile's life, in every second tile
    // (HEAD_LAST: the one-level kernels, cfg2 chain +0.4 %; level 2 keeps thread 0's extra job -- there it measured -0.4 % at cfg5, r4ac)
    a.head_owned = HEAD_LAST && a.pairs && a.jh > a.ja && a.jh + 2ull * (NT - 1) >= a.jb;          // (wave-uniform)
    if (a.head_owned && tid == NT - 1) { a.j0 = a.ja; a.npts = 1; }
    const unsigned ph0 = p.phi0 + (unsigned)a.j0 * p.step, ph1 = ph0 + p.step;      // low 32 bits are all the bank index needs
    a.b0p = p.arb_bank + (a.npts >= 1 ? (ph0 & 0xffffffu) >> 16 : 0u) * 14u;
    a.b1p = p.arb_bank + (a.npts >= 2 ? (ph1 & 0xffffffu) >> 16 : 0u) * 14u;
    return a;
}

template <int NT, int GS>
static __device__ __forceinline__ void fe_arb_store(const pmr_fe_params &p, const fe_arb_plan &a, unsigned long long qa,
                                                    const cf *fin, const float (&bk0)[14], const float (&bk1)[14], int tid)
{
    cf *__restrict__ out = (cf *)p.out;
    const auto resamp = [&](unsigned long long j, const float *bk) {
        const unsigned long long ph = (unsigned long long)p.phi0 + j * p.step;
        const int qd = (int)((ph >> 24) - qa) + p.HhQ;                            // tile-local decimated index
        const int ql = qd - 13;
        // window sample ql + k sits at (ql + k) + ((ql + k) >> GS) (layout L(1 << GS)).  With ql = Q 2^GS + r and k = K 2^GS + g that is
        // [ql + Q + ((r + g) >> GS)] + [k + K]: one base per residue g, every tap a compile-time offset -- three address
        // instructions per tap otherwise, more than the tap's own multiply-add
        constexpr int NG = 1 << GS;
        const cf *bg[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) bg[g] = fin + ql + (ql >> GS) + (((ql & (NG - 1)) + g) >> GS);
        cf y = cfm(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 14; k++) y = cfma(bk[k], bg[k & (NG - 1)][k + (k >> GS)], y);
        return y;
    };
    if (a.npts == 2) {
        const cf y0 = resamp(a.j0, bk0), y1 = resamp(a.j0 + 1, bk1);
        FE_STORE4(out + ((p.out_pos0 + a.j0) & (p.out_mask & FE_OUT_AND)), make_float4(y0.x, y0.y, y1.x, y1.y));
    } else if (a.npts == 1) {
        out[(p.out_pos0 + a.j0) & (p.out_mask & FE_OUT_AND)] = resamp(a.j0, bk0);
    }
    // rare leftovers, taps fetched on the spot: the odd head sample of a tile whose last thread is busy, and anything beyond 2 NT
    // outputs per tile
    if (tid == 0 && a.jh > a.ja && a.ja < a.jb && !a.head_owned) {
        const unsigned long long ph = (unsigned long long)p.phi0 + a.ja * p.step;
        out[(p.out_pos0 + a.ja) & (p.out_mask & FE_OUT_AND)] = resamp(a.ja, p.arb_bank + ((unsigned)(ph & 0xffffffu) >> 16) * 14u);
    }
    for (unsigned long long j = (a.pairs ? a.jh + 2ull * NT : a.ja + NT) + tid; j < a.jb; j += NT) {
        const unsigned long long ph = (unsigned long long)p.phi0 + j * p.step;
        out[(p.out_pos0 + j) & (p.out_mask & FE_OUT_AND)] = resamp(j, p.arb_bank + ((unsigned)(ph & 0xffffffu) >> 16) * 14u);
    }
}


// launch with optional per-launch events (pmr_launch_events, pmr_kernels.h)
#define PMR_LAUNCH_EV(kern, grid, block, lds, st, ev, ...)                                                                        \
    do {                                                                                                                           \
        if ((ev) && ((ev)->start || (ev)->stop)) {                                                                                 \
            if (pmr_debug_poison_enabled()) (void)pmr_debug_poison_lds((pmr_stream_t)(st));                                        \
            hipExtLaunchKernelGGL(kern, grid, block, (unsigned)(lds), st, (hipEvent_t)(ev)->start, (hipEvent_t)(ev)->stop, 0,      \
                                  __VA_ARGS__);                                                                                    \
        } else PMR_KLAUNCH(kern, grid, block, lds, st, __VA_ARGS__);                                                          \
    } while (0)

#endif
