// pmr_frontend.hip -- FUSED front end for gfx950: dc-block -> half-band cascade -> arbitrary resampler
// (reference src/sdr_pmr446.c:795-796) in ONE pass over the raw cf32 block.
//
// HBM traffic: every raw sample is read once (8 B), plus a tile halo that is normally L2-resident;
// only the resampled stream (8*rate B per input sample) is written.  Everything else lives in LDS.
//
// Tiling.  Workgroup c owns decimated samples [c*TQ, (c+1)*TQ) and computes them from raw samples
// [c*T_own - Hh, (c+1)*T_own) of the "raw_rel" axis (origin = first sample of the first new decimation
// group, i.e. `pend` samples before the new block; negative block indices come from the raw history the
// chain keeps).  The tile holds N0 = 16*NT samples; Hh >= sum of all filter histories is recomputed.
//
// DC blocker without a serial dependency.  v[n] = x[n] + lambda v[n-1] is a linear scan; inside the tile it
// runs from ZERO state (thread-serial over 16 samples, wave shuffle scan, cross-wave Horner).  The missing
// carry-in V_c = v[tile start - 1] only adds  alpha*V_c*lambda^r  to yb, an exponential, and exponentials
// are eigenfunctions of every later (linear) stage.  So the tile also records two probes of its local scan,
// a tiny kernel (k_fe_tiles) turns the probes of all tiles into the V_c, and k_fe_dcfix subtracts
// V_c * K * mu^q' * GA[idx] from the resampled samples, where K, mu, GA are the closed-form gains of the
// cascade for that exponential (tests/chain_model.py documents the sums; the identity is exact in exact
// arithmetic and was checked to 1e-15 in float64).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pmr_kernels.h"

// hipFuncSetAttribute is per device: a process may hold handles on several GPUs (pmr_chain_cfg.device), so the
// "already raised the dynamic-LDS limit" flag is one bit per device ordinal, not one bool per process
static inline bool pmr_attr_needed(unsigned long long &mask)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    if (mask >> dev & 1ull) return false;
    mask |= 1ull << dev;
    return true;
}

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
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
static __device__ __forceinline__ cf dpp0c(cf v) { return cf{dpp0<CTRL, ROW_MASK>(v.x), dpp0<CTRL, ROW_MASK>(v.y)}; }

// LDS layout L(G): element e lives at e + e/G (one 8-byte pad per G elements) so that threads whose chunks
// are G elements apart hit distinct banks with ds_read_b64 / ds_write_b64 (stride 2G+2 dwords, gcd with 64 = 2).
template <int G> static __device__ __forceinline__ int lidx(int e) { return e + e / G; }
static __device__ __forceinline__ int lidx_rt(int e, int g_shift) { return e + (e >> g_shift); }

// ---------------------------------------------------------------------------------------------
// one half-band stage out of LDS, P outputs per thread, 2*MM branch taps   
//   z1[o] = z0[2o+1-2m] + sum_j h1[j] * z0[2o - 2(2m-1-j)]
// reads layout L(2P), writes layout L(max(P,2)) in place (two barriers).  Every LDS address is
// thread_base + compile-time constant (floor division keeps that true left of the tile, where the
// reads land in the zero pad in front of the buffer and only feed outputs inside the halo).
// ---------------------------------------------------------------------------------------------
#ifndef FE_WAVES_512
#define FE_WAVES_512 8      /* waves per SIMD requested for the 512 x 8 geometry (4 workgroups = 32 waves per CU) */
#endif
#ifndef FE_WAVES_256x8
#define FE_WAVES_256x8 6     /* waves per SIMD requested for the 256 x 8 geometry (2048-sample tiles) */
#endif
#define FE_PAD 64      /* elements in front of the tile buffer; >= (4*10-2) * (1 + 1/2) */

static constexpr int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
template <int G> static constexpr int loff(int e) { return e + fdiv(e, G); }   // layout offset of a constant index

template <int P>
static __device__ __forceinline__ void hb_store(cf *buf, int tid, int n_threads, const cf (&y)[P])
{
    __syncthreads();
    if (tid < n_threads) {
        if constexpr (P >= 2) {
            cf *o = buf + tid * (P + 1);               // L(P): element tid*P + p at tid*(P+1) + p
#pragma unroll
            for (int p = 0; p < P; p++) o[p] = y[p];
        } else {
            buf[tid + (tid >> 1)] = y[0];              // L(2)
        }
    }
    __syncthreads();
}

template <int P, int MM>
static __device__ __forceinline__ void hb_stage(cf *buf, int tid, int n_threads, const float *__restrict__ h1,
                                                float scale)
{
    constexpr int NE = P + 2 * MM - 1;             // even-offset window elements
    constexpr int G = 2 * P;
    cf y[P];
    if (tid < n_threads) {
        const cf *w = buf + tid * (G + 1);         // thread chunk base: element 2*tid*P
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
    hb_store<P>(buf, tid, n_threads, y);
}

// fallback for other half-band lengths: taps looped, window read straight from LDS
template <int P>
static __device__ void hb_stage_generic(cf *buf, int tid, int n_threads, int mm, const float *__restrict__ h1,
                                        float scale)
{
    constexpr int G = 2 * P;
    cf y[P];
    if (tid < n_threads) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int o = tid * P + p;
            cf a = cfm(0.f, 0.f);
            for (int j = 0; j < 2 * mm; j++) {
                int e = 2 * o - 2 * (2 * mm - 1 - j);
                e = e < 0 ? 0 : e;
                a = cfma(h1[j], buf[lidx<G>(e)], a);
            }
            int e = 2 * o + 1 - 2 * mm;
            e = e < 0 ? 0 : e;
            y[p] = cadd_scale(buf[lidx<G>(e)], a, scale);
        }
    }
    hb_store<P>(buf, tid, n_threads, y);
}

template <int P>
static __device__ __forceinline__ void hb_dispatch(cf *buf, int tid, int n_threads, int mm,
                                                   const float *__restrict__ h1, float scale)
{
    // register-window variants for the (P, m) pairs the As = 60 dB cascade produces (m = 3,..,3,5,10 in
    // execution order: long filters only ever meet small P, except in 1- and 2-stage cascades)
    if (mm == 3)                 hb_stage<P, 3>(buf, tid, n_threads, h1, scale);
    else if (mm == 5)            hb_stage<P, 5>(buf, tid, n_threads, h1, scale);
    else if (mm == 10 && P <= 2) hb_stage<(P <= 2 ? P : 1), 10>(buf, tid, n_threads, h1, scale);
    else                         hb_stage_generic<P>(buf, tid, n_threads, mm, h1, scale);
}

// ---------------------------------------------------------------------------------------------
// exact ceil(num / den) for num < 2^58, den < 2^26 without the 64-bit integer division routine:
// double-precision estimate, then an integer fix-up
static __device__ __forceinline__ unsigned long long ceil_div_u64(unsigned long long num, unsigned den)
{
    unsigned long long q = (unsigned long long)((double)num / (double)den);
    while (q * den < num) q++;
    while (q > 0 && (q - 1) * den >= num) q--;
    return q;
}

// NT threads own a tile of N0 = NT * SPT input samples (SPT consecutive samples per thread in the dc scan).
// MODE 0: whole front end (raw in -> dc-block -> all stages -> resampler).
// Deep cascades (tile halo would eat the tile) run as two launches of the same code:
// MODE 1: level 1 = raw in -> dc-block -> the first few (6-tap) stages -> decimated stream stored to a ring;
// MODE 2: level 2 = ring in (dc carry of level 1 applied at load) -> remaining stages -> resampler.
enum { FE_FULL = 0, FE_L1 = 1, FE_L2 = 2 };

template <int NT, int SPT, int MODE>
__global__ __launch_bounds__(NT, (NT == 512 && SPT == 8) ? FE_WAVES_512 : (NT == 256 && SPT == 8) ? FE_WAVES_256x8 : 4) void k_frontend(pmr_fe_params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int N0 = NT * SPT;
    constexpr int LPT = N0 / 2 / NT;                        // 16-byte loads per thread in phase A
    cf *buf = reinterpret_cast<cf *>(smem) + FE_PAD;        // [FE_PAD zero pad | N0 + N0/SPT elements | scan scratch]
    cf *wagg = buf + (N0 + N0 / SPT);                       // [NT/64] wave aggregates of the dc scan

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const cf *__restrict__ x = (const cf *)p.x;
    const cf *__restrict__ hist = (const cf *)p.hist;
    const float lam = -p.dc_a1;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so give every XCD
    // a CONTIGUOUS range of tiles -- a tile's halo is its left neighbour's tail and can then be an L2 hit instead of a
    // second HBM read (PMC: 8 % extra fetch without it).  Placement only affects speed, never results.
    int c = blockIdx.x;
    {
        const int nt_all = gridDim.x, per = nt_all >> 3, main = per << 3;
        if (c < main) c = (c & 7) * per + (c >> 3);
    }
    const long b0 = (long)c * p.T_own - p.Hh - p.pend;      // block-relative index of tile sample 0
    unsigned long long *stamps = (unsigned long long *)p.stamps;   // diagnostic build only (PMR_FE_STAMP)
    long long ts[5] = {0, 0, 0, 0, 0};
    if (stamps) ts[0] = clock64();

    // ---- resampler bookkeeping first: which outputs this tile owns, and the polyphase taps of this thread's first
    // output (a dependent global load: issued now, it lands while the raw tile streams in) ----
    const unsigned long long qa = (unsigned long long)c * p.TQ;
    unsigned long long ja = 0, jb = 0;
    {
        unsigned long long qb = qa + p.TQ;
        if (qb > p.Q) qb = p.Q;
        if (MODE != FE_L1 && qa < qb && !(p.ablate & 8)) {
            const unsigned long long sa = qa << 24, sb = qb << 24;
            ja = sa <= p.phi0 ? 0ull : ceil_div_u64(sa - p.phi0, p.step);
            jb = sb <= p.phi0 ? 0ull : ceil_div_u64(sb - p.phi0, p.step);
            if (jb > p.ny) jb = p.ny;
        }
        if (p.tile_j && tid == 0 && MODE != FE_L1) { ((unsigned long long *)p.tile_j)[2 * c] = ja; ((unsigned long long *)p.tile_j)[2 * c + 1] = jb; }
    }
    float bk0[14];
    {
        const unsigned long long j = ja + tid;
        const unsigned long long ph = (unsigned long long)p.phi0 + j * p.step;
        const unsigned idx = j < jb ? (unsigned)(ph & 0xffffffu) >> 16 : 0u;
        const float *b = p.arb_bank + idx * 14u;
#pragma unroll
        for (int k = 0; k < 14; k++) bk0[k] = b[k];
    }

    // ---- phase A: raw samples -> LDS (layout L(SPT)); history for b < 0, zeros beyond the block ----
    if (tid < FE_PAD) buf[tid - FE_PAD] = cfm(0.f, 0.f);
    if (MODE == FE_L2) {
        // level-2 input: the decimated ring written by level 1; sample r of the tile has absolute index in_abs0 + b0 + r.
        // Samples produced by THIS call (abs >= in_abs0) still miss their dc carry: V_c1 * K1 * mu^i' (see k_fe_dcfix).
        const cf *__restrict__ ring = (const cf *)p.in_ring;
        const cf *__restrict__ V1 = (const cf *)p.fixV;
        for (int i = tid; i < N0; i += NT) {
            const long long a = (long long)p.in_abs0 + b0 + i;
            const long long jn = b0 + i;                              // index among this call's new samples
            cf v = cfm(0.f, 0.f);
            if (a >= 0 && jn < (long long)p.n_in) {
                v = ring[(unsigned long long)a & p.in_mask];
                if (V1 && jn >= 0) {
                    const unsigned c1 = (unsigned)jn / p.fix_TQ;
                    const unsigned ql = (unsigned)jn - c1 * p.fix_TQ + p.fix_HhQ;
                    const float g = p.fix_K * (p.fix_T1[ql >> 5] * p.fix_T2[ql & 31]);
                    const cf Vc = V1[c1];
                    v = cf{fmaf(-Vc.x, g, v.x), fmaf(-Vc.y, g, v.y)};
                }
            }
            buf[lidx<SPT>(i)] = v;
        }
    } else if (!(p.ablate & 1)) {
        // interior tile, 16-byte aligned: all loads of the thread are issued before the first LDS write
        const bool fast = b0 >= 0 && b0 + N0 <= (long)p.n_in && ((reinterpret_cast<uintptr_t>(x + b0) & 15) == 0);
        if (fast) {
            const float4 *__restrict__ src = reinterpret_cast<const float4 *>(x + b0);
            float4 v[LPT];
#pragma unroll
            for (int i = 0; i < LPT; i++) v[i] = src[tid + NT * i];
#pragma unroll
            for (int i = 0; i < LPT; i++) {
                cf *d = buf + lidx<SPT>(2 * (tid + NT * i));      // the pair never straddles an SPT-sample chunk
                d[0] = cfm(v[i].x, v[i].y);
                d[1] = cfm(v[i].z, v[i].w);
            }
        } else {
#pragma unroll 4
            for (int i = tid; i < N0; i += NT) {
                const long b = b0 + i;
                cf v = cfm(0.f, 0.f);
                if (b < 0) { const long hi = (long)p.hcap + b; if (hi >= 0) v = hist[hi]; }
                else if (b < (long)p.n_in) v = x[b];
                buf[lidx<SPT>(i)] = v;
            }
        }
    }
    __syncthreads();
    if (stamps) ts[1] = clock64();

    // ---- phase B: dc blocker (:795) from zero state.  With a 6-tap first stage (m = 3, every cascade of >= 3 stages)
    // the first half-band stage is computed right here from registers: no write-back of yb, no window re-read. ----
    cf *bnd = wagg + NT / 64;                                // [NT/64][10] last 10 dc-blocked samples of every wave
    const bool fuse0 = SPT == 16 && p.h >= 1 && p.m[0] == 3 && !(p.ablate & 6);
    if (!(p.ablate & 2)) {
        const float lp = p.lam_lane_pow[lane];              // lambda^(SPT lane)
        cf xs[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++) xs[j] = buf[(SPT + 1) * tid + j];
        cf yb[SPT];
        if constexpr (MODE == FE_L2) {
            (void)lp;
#pragma unroll
            for (int j = 0; j < SPT; j++) yb[j] = xs[j];                   // level 2: the input is already dc-blocked
        } else {
        cf v = cfm(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SPT; j++) v = cfma(lam, v, xs[j]);             // v0 = x - a1 v1
        // inclusive decayed scan across the wave (DPP): inc_l = sum_{s<=l} lambda^(SPT (l-s)) agg_s
        v = cfma(p.lam_pow16[0], dpp0c<0x111>(v), v);                      // row_shr:1
        v = cfma(p.lam_pow16[1], dpp0c<0x112>(v), v);                      // row_shr:2
        v = cfma(p.lam_pow16[2], dpp0c<0x114>(v), v);                      // row_shr:4
        v = cfma(p.lam_pow16[3], dpp0c<0x118>(v), v);                      // row_shr:8
        v = cfma(p.lam_lane_pow[(lane & 15) + 1], dpp0c<0x142, 0xA>(v), v);   // row_bcast:15 -> rows 1, 3
        v = cfma(p.lam_lane_pow[(lane & 31) + 1], dpp0c<0x143, 0xC>(v), v);   // row_bcast:31 -> rows 2, 3
        if (lane == 63) wagg[wave] = v;
        const cf ex = dpp0c<0x138>(v);                                     // wave_shr:1 (lane 0 <- 0)
        __syncthreads();
        cf cw = cfm(0.f, 0.f);                                             // v (local) at the end of the previous wave
        for (int w = 0; w < wave; w++) cw = cfma(p.lam_wave, cw, wagg[w]);
        cf v1 = cfma(lp, cw, ex);
        // stray probes (block start - 1 in tile 0, block end in the last tile) sit at arbitrary offsets
        const int pL = (c == 0) ? p.Hh + p.pend - 1 : -1;
        const int pE = (c == p.c_end) ? p.off_end : -1;
        const bool stray = (pL >= 0 && pL / SPT == tid) || (pE >= 0 && pE / SPT == tid);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const cf v0 = cfma(lam, v1, xs[j]);
            yb[j] = csub(v0, v1);                                          // y = v0 - v1
            v1 = v0;
            if (stray) {
                if (SPT * tid + j == pL) ((cf *)p.probeL)[0] = v0;
                if (SPT * tid + j == pE) ((cf *)p.probeE)[0] = v0;
            }
        }
        if (tid == p.Hh / SPT - 1) ((cf *)p.probeA)[c] = v1;               // local v at tile offset Hh-1
        if (tid == NT - 1) ((cf *)p.probeB)[c] = v1;                       // local v at tile offset N0-1
        }
        if constexpr (SPT == 16) {
            if (fuse0) {
                // halo of stage 0: the previous thread's yb[6..15] (lane 0: previous wave's lane 63, through LDS)
                cf W[26];
#pragma unroll
                for (int i = 0; i < 10; i++) W[i] = dpp0c<0x138>(yb[6 + i]);
#pragma unroll
                for (int i = 0; i < 16; i++) W[10 + i] = yb[i];
                if (lane == 63) {
#pragma unroll
                    for (int i = 0; i < 10; i++) bnd[wave * 10 + i] = yb[6 + i];
                }
                __syncthreads();                                           // also: every thread is done with the raw tile
                if (lane == 0 && wave > 0) {
#pragma unroll
                    for (int i = 0; i < 10; i++) W[i] = bnd[(wave - 1) * 10 + i];
                }
                // z1[8 tid + q] = W[2q + 5] + sum_j h1[j] W[2q + 2j]   (window offset 0 <-> sample 16 tid - 10)
                const float *__restrict__ h1 = p.hb_taps + p.tap_off[0];
                const float scale0 = p.h == 1 ? p.zeta : 1.0f;
                cf *o = buf + tid * 9;                                     // layout L(8)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    cf a = cfm(0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < 6; j++) a = cfma(h1[j], W[2 * q + 2 * j], a);
                    o[q] = cadd_scale(W[2 * q + 5], a, scale0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < SPT; j++) buf[(SPT + 1) * tid + j] = yb[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < SPT; j++) buf[(SPT + 1) * tid + j] = yb[j];
        }
    }
    __syncthreads();
    if (stamps) ts[2] = clock64();

    // ---- phase C: half-band cascade, in place (stage e halves the sample count) ----
    int g_shift = SPT == 16 ? 4 : 3;                         // layout of the current signal: L(1 << g_shift)
    if (!(p.ablate & 4)) {
        const float *__restrict__ taps = p.hb_taps;
        int n_out = N0 >> 1;
        int e0 = 0;
        if (fuse0) { e0 = 1; n_out >>= 1; g_shift = 3; }     // stage 0 already done, its output sits in L(8)
        for (int e = e0; e < p.h; e++) {
            const int mm = p.m[e];
            const float *h1 = taps + p.tap_off[e];
            const float scale = (e == p.h - 1) ? p.zeta : 1.0f;
            const int pp = n_out >= NT ? n_out / NT : 1;     // outputs per thread: SPT/2, ..., 2, 1, 1, ...
            const int n_threads = n_out / pp;
            if (SPT >= 16 && pp == 8) hb_dispatch<8>(buf, tid, n_threads, mm, h1, scale);
            else if (pp == 4)         hb_dispatch<4>(buf, tid, n_threads, mm, h1, scale);
            else if (pp == 2)         hb_dispatch<2>(buf, tid, n_threads, mm, h1, scale);
            else                      hb_dispatch<1>(buf, tid, n_threads, mm, h1, scale);
            g_shift = pp == 8 ? 3 : (pp == 4 ? 2 : 1);       // output layout L(max(pp, 2))
            n_out >>= 1;
        }
    }
    if (stamps) ts[3] = clock64();

    // ---- level 1: store the owned part of the last stage's output to the decimated ring; no resampler ----
    if constexpr (MODE == FE_L1) {
        (void)bk0; (void)ja; (void)jb;
        cf *__restrict__ out = (cf *)p.out;
        const unsigned long long q0 = qa;                               // first decimated sample this tile owns
        for (int i = tid; i < p.TQ; i += NT) {
            const unsigned long long q1 = q0 + i;
            if (q1 < p.Q) out[(p.out_pos0 + q1) & p.out_mask] = buf[lidx_rt(p.HhQ + i, g_shift)];
        }
    }
    // ---- phase D: arbitrary resampler (24-bit phase) for the outputs whose input sample is owned here ----
    if (MODE != FE_L1 && !(p.ablate & 8)) {
        cf *__restrict__ out = (cf *)p.out;
        bool first = true;
        for (unsigned long long j = ja + tid; j < jb; j += NT) {
            const unsigned long long ph = (unsigned long long)p.phi0 + j * p.step;
            const int ql = (int)((ph >> 24) - qa) + p.HhQ;              // tile-local decimated index
            float bk[14];
            if (first) {
#pragma unroll
                for (int k = 0; k < 14; k++) bk[k] = bk0[k];
            } else {
                const unsigned idx = (unsigned)(ph & 0xffffffu) >> 16;
                const float *b = p.arb_bank + idx * 14u;
#pragma unroll
                for (int k = 0; k < 14; k++) bk[k] = b[k];
            }
            first = false;
            cf y = cfm(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 14; k++) y = cfma(bk[k], buf[lidx_rt(ql - 13 + k, g_shift)], y);
            out[(p.out_pos0 + j) & p.out_mask] = y;
        }
    }
    // ---- raw history for the next call (last hcap samples of old history || block), by workgroup 0 ----
    if (MODE != FE_L2 && c == 0 && p.new_hist) {
        cf *__restrict__ nh = (cf *)p.new_hist;
        for (int i = tid; i < p.hcap; i += NT) {
            const long sb = (long)i + (long)p.n_in - (long)p.hcap;      // block-relative index
            nh[i] = sb < 0 ? hist[(long)i + p.n_in] : x[sb];
        }
    }
    if (stamps && tid == 0) {
        ts[4] = clock64();
        atomicAdd(&stamps[0], (unsigned long long)(ts[1] - ts[0]));   // A: load + LDS write
        atomicAdd(&stamps[1], (unsigned long long)(ts[2] - ts[1]));   // B: dc scan
        atomicAdd(&stamps[2], (unsigned long long)(ts[3] - ts[2]));   // C: cascade
        atomicAdd(&stamps[3], (unsigned long long)(ts[4] - ts[3]));   // D: resampler
        atomicAdd(&stamps[4], 1ull);
    }
}

// ---------------------------------------------------------------------------------------------
// SPECIALISED front end for the cascades the As = 60 dB design produces: N3 six-tap stages (m = 3), optionally followed
// by the m = 5 and m = 10 stages (TAIL), 256 threads x 16 samples.  Same arithmetic as k_frontend<256, 16, MODE>, same
// order of operations; what changes is when things are fetched and how often the workgroup synchronises (the kernel is
// latency-bound: its throughput is (tiles in flight) / (time one tile spends waiting)):
//   * every table the tile needs -- branch taps (kernel-argument segment -> SGPRs), lambda powers, the polyphase taps of
//     BOTH resampler outputs a thread can own -- is requested before the raw tile, not at the point of use behind a barrier;
//   * the cascade ping-pongs between two LDS regions (z1 -> R0, z2 -> R1, z3 -> R0, ...), so a stage is
//     read -> compute -> write -> ONE barrier instead of read -> barrier -> write -> barrier;
//   * stage count, per-stage outputs per thread and LDS layouts are compile-time, so every LDS address is
//     thread base + immediate.
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

template <int MODE, int N3, int TAIL>
__global__ __launch_bounds__(256, MODE == FE_L2 ? 4 : 5) void k_frontend_fast(pmr_fe_params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256, SPT = 16, N0 = NT * SPT, LPT = N0 / 2 / NT;
    constexpr int H = N3 + 2 * TAIL;
    constexpr int R1_OFF = (N0 / 2) + (N0 / 2) / 8;         // z1 (2048 samples, layout L(8)) fills [0, R1_OFF) of the tile
    // LDS budget.  The raw tile never sits in LDS as a whole: it passes through in two halves (2048 samples each, layout
    // L(16)) on its way into the threads' registers, in the area z1 will occupy afterwards.  What stays is z1 (R0, 2304
    // slots) + z2 (R1, 1280 slots) + scratch = 29.5 KB, i.e. FIVE tiles per CU instead of the four a whole raw tile
    // (34.8 KB) allows -- this kernel's throughput is proportional to the tiles in flight.  (Level 2 keeps the whole-tile
    // staging: its input is 1/8 of the data.)
    constexpr int HALF = N0 / 2;
    constexpr int BODY = MODE == FE_L2 ? (N0 + N0 / SPT) : (R1_OFF + (N0 / 4) + (N0 / 4) / 4);   // R0 + R1
    cf *buf = reinterpret_cast<cf *>(smem) + FE_PAD;
    cf *wagg = buf + BODY;
    cf *bnd = wagg + NT / 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const cf *__restrict__ x = (const cf *)p.x;
    const cf *__restrict__ hist = (const cf *)p.hist;
    const float lam = -p.dc_a1;
    int c = blockIdx.x;
    {
        const int nt_all = gridDim.x, per = nt_all >> 3, main = per << 3;
        if (c < main) c = (c & 7) * per + (c >> 3);         // XCD-aware tile order (see k_frontend)
    }
    const long b0 = (long)c * p.T_own - p.Hh - p.pend;

    // ---- everything that comes from tables, requested up front ----
    const unsigned long long qa = (unsigned long long)c * p.TQ;
    unsigned long long ja = 0, jb = 0, jh = 0;
    bool pairs = false;
    float bk0[14], bk1[14];
    const float *b0p = nullptr, *b1p = nullptr;
    if constexpr (MODE != FE_L1) {
        unsigned long long qb = qa + p.TQ;
        if (qb > p.Q) qb = p.Q;
        if (qa < qb) {
            const unsigned long long sa = qa << 24, sb = qb << 24;
            ja = sa <= p.phi0 ? 0ull : ceil_div_u64(sa - p.phi0, p.step);
            jb = sb <= p.phi0 ? 0ull : ceil_div_u64(sb - p.phi0, p.step);
            if (jb > p.ny) jb = p.ny;
        }
        if (p.tile_j && tid == 0) { ((unsigned long long *)p.tile_j)[2 * c] = ja; ((unsigned long long *)p.tile_j)[2 * c + 1] = jb; }
        // a thread owns the PAIR of adjacent outputs jh + 2 tid, + 1 (jh = ja rounded up to an even ring position; the
        // odd head sample, if any, is thread 0's extra job): one 16-byte store per thread instead of two 8-byte ones
        // (only when the tile has more outputs than threads -- otherwise one output per thread is the shorter phase D)
        pairs = jb - ja > (unsigned long long)NT;
        jh = pairs ? ja + ((p.out_pos0 + ja) & 1ull) : ja;
        const unsigned long long j0 = pairs ? jh + 2ull * tid : ja + tid, j1 = pairs ? j0 + 1 : jb;
        const unsigned long long ph0 = (unsigned long long)p.phi0 + j0 * p.step, ph1 = ph0 + p.step;
        b0p = p.arb_bank + (j0 < jb ? (unsigned)(ph0 & 0xffffffu) >> 16 : 0u) * 14u;
        b1p = p.arb_bank + (j1 < jb ? (unsigned)(ph1 & 0xffffffu) >> 16 : 0u) * 14u;
    }
    float lp = 0.f, l15 = 0.f, l31 = 0.f;
    if constexpr (MODE != FE_L2) {
        lp = p.lam_lane_pow[lane]; l15 = p.lam_lane_pow[(lane & 15) + 1]; l31 = p.lam_lane_pow[(lane & 31) + 1];
    }

    // ---- phase A: raw samples -> LDS (layout L(16)) ----
    if (tid < FE_PAD) buf[tid - FE_PAD] = cfm(0.f, 0.f);
    if constexpr (MODE == FE_L2) {
        const cf *__restrict__ ring = (const cf *)p.in_ring;
        const cf *__restrict__ V1 = (const cf *)p.fixV;
        // Two batches of eight samples per thread: the ring loads of a batch are all issued first, then the table
        // look-ups of level 1's dc carry (tile index and local offset advance incrementally: one division per thread), then
        // the arithmetic -- a one-sample-per-iteration loop pays the load latency sixteen times in a row.
        unsigned c1 = 0, ql = 0; bool trk = false;
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {
            cf v[8], Vc[8]; float g[8]; bool fx[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = tid + NT * (8 * hb + k);
                const long long a = (long long)p.in_abs0 + b0 + i, jn = b0 + i;
                v[k] = cfm(0.f, 0.f);
                if (a >= 0 && jn < (long long)p.n_in) v[k] = ring[(unsigned long long)a & p.in_mask];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = tid + NT * (8 * hb + k);
                const long long a = (long long)p.in_abs0 + b0 + i, jn = b0 + i;
                fx[k] = V1 && a >= 0 && jn >= 0 && jn < (long long)p.n_in;
                g[k] = 0.f; Vc[k] = cfm(0.f, 0.f);
                if (fx[k]) {
                    if (!trk) { c1 = (unsigned)jn / p.fix_TQ; ql = (unsigned)jn - c1 * p.fix_TQ; trk = true; }
                    else { ql += NT; while (ql >= p.fix_TQ) { ql -= p.fix_TQ; c1++; } }
                    const unsigned e = ql + p.fix_HhQ;
                    g[k] = p.fix_K * (p.fix_T1[e >> 5] * p.fix_T2[e & 31]);
                    Vc[k] = V1[c1];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = tid + NT * (8 * hb + k);
                cf w = v[k];
                if (fx[k]) w = cf{fmaf(-Vc[k].x, g[k], w.x), fmaf(-Vc[k].y, g[k], w.y)};
                buf[lidx<SPT>(i)] = w;
            }
        }
    }
    cf xs[SPT];
    if constexpr (MODE == FE_L2) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) xs[j] = buf[(SPT + 1) * tid + j];
    } else {
        const bool fast = b0 >= 0 && b0 + N0 <= (long)p.n_in && ((reinterpret_cast<uintptr_t>(x + b0) & 15) == 0);
        float4 v[LPT];
        if (fast) {
            const float4 *__restrict__ src = reinterpret_cast<const float4 *>(x + b0);
#pragma unroll
            for (int i = 0; i < LPT; i++) v[i] = src[tid + NT * i];       // all eight loads in flight before the first LDS write
        }
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            if (fast) {
#pragma unroll
                for (int i = 0; i < LPT / 2; i++) {
                    cf *d = buf + lidx<SPT>(2 * (tid + NT * i));          // local index inside the half
                    const float4 w = v[hf * (LPT / 2) + i];
                    d[0] = cfm(w.x, w.y);
                    d[1] = cfm(w.z, w.w);
                }
            } else {
#pragma unroll 4
                for (int i = tid; i < HALF; i += NT) {
                    const long b = b0 + hf * HALF + i;
                    cf w = cfm(0.f, 0.f);
                    if (b < 0) { const long hi = (long)p.hcap + b; if (hi >= 0) w = hist[hi]; }
                    else if (b < (long)p.n_in) w = x[b];
                    buf[lidx<SPT>(i)] = w;
                }
            }
            __syncthreads();
            if ((tid >> 7) == hf) {                                       // threads 0..127 own the first half, 128..255 the second
#pragma unroll
                for (int j = 0; j < SPT; j++) xs[j] = buf[(SPT + 1) * (tid & 127) + j];
            }
            if (hf == 0) __syncthreads();                                 // the second half overwrites the staging area
        }
    }

    // polyphase taps of the (at most two) resampler outputs this thread owns: requested now that the raw tile has left
    // the registers it was loaded into, consumed in phase D -- phases B and C hide the latency
    if constexpr (MODE != FE_L1) {
#pragma unroll
        for (int k = 0; k < 14; k++) { bk0[k] = b0p[k]; bk1[k] = b1p[k]; }
    }

    // ---- phase B: dc blocker from zero state + first (six-tap) stage from registers ----
    {
        cf yb[SPT];
        if constexpr (MODE == FE_L2) {
#pragma unroll
            for (int j = 0; j < SPT; j++) yb[j] = xs[j];
        } else {
            cf v = cfm(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < SPT; j++) v = cfma(lam, v, xs[j]);
            v = cfma(p.lam_pow16[0], dpp0c<0x111>(v), v);
            v = cfma(p.lam_pow16[1], dpp0c<0x112>(v), v);
            v = cfma(p.lam_pow16[2], dpp0c<0x114>(v), v);
            v = cfma(p.lam_pow16[3], dpp0c<0x118>(v), v);
            v = cfma(l15, dpp0c<0x142, 0xA>(v), v);
            v = cfma(l31, dpp0c<0x143, 0xC>(v), v);
            if (lane == 63) wagg[wave] = v;
            const cf ex = dpp0c<0x138>(v);
            __syncthreads();
            cf cw = cfm(0.f, 0.f);
            for (int w = 0; w < wave; w++) cw = cfma(p.lam_wave, cw, wagg[w]);
            cf v1 = cfma(lp, cw, ex);
            const int pL = (c == 0) ? p.Hh + p.pend - 1 : -1;
            const int pE = (c == p.c_end) ? p.off_end : -1;
            const bool stray = (pL >= 0 && pL / SPT == tid) || (pE >= 0 && pE / SPT == tid);
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const cf v0 = cfma(lam, v1, xs[j]);
                yb[j] = csub(v0, v1);
                v1 = v0;
                if (stray) {
                    if (SPT * tid + j == pL) ((cf *)p.probeL)[0] = v0;
                    if (SPT * tid + j == pE) ((cf *)p.probeE)[0] = v0;
                }
            }
            if (tid == p.Hh / SPT - 1) ((cf *)p.probeA)[c] = v1;
            if (tid == NT - 1) ((cf *)p.probeB)[c] = v1;
        }
        cf W[26];
#pragma unroll
        for (int i = 0; i < 10; i++) W[i] = dpp0c<0x138>(yb[6 + i]);
#pragma unroll
        for (int i = 0; i < 16; i++) W[10 + i] = yb[i];
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < 10; i++) bnd[wave * 10 + i] = yb[6 + i];
        }
        __syncthreads();                                   // also: every thread is done with the raw tile
        if (lane == 0 && wave > 0) {
#pragma unroll
            for (int i = 0; i < 10; i++) W[i] = bnd[(wave - 1) * 10 + i];
        }
        const float scale0 = H == 1 ? p.zeta : 1.0f;
        cf *o = buf + tid * 9;                             // z1 in layout L(8), region R0
#pragma unroll
        for (int q = 0; q < 8; q++) {
            cf a = cfm(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 6; j++) a = cfma(p.taps_k[j], W[2 * q + 2 * j], a);
            o[q] = cadd_scale(W[2 * q + 5], a, scale0);
        }
    }
    __syncthreads();

    // ---- phase C: remaining stages, ping-pong R0 <-> R1; stage e (execution index) has 2048 >> e outputs ----
    cf *R0 = buf, *R1 = buf + R1_OFF;
#define FE_STAGE(E, MM, TOFF) do { constexpr int NOUT = (N0 / 2) >> (E); constexpr int PP = NOUT >= NT ? NOUT / NT : 1;          \
        hb_stage_pp<PP, MM>(((E) & 1) ? R0 : R1, ((E) & 1) ? R1 : R0, tid, NOUT / PP, p.taps_k + (TOFF),                         \
                            (E) == H - 1 ? p.zeta : 1.0f); } while (0)
    // six-tap stages 1 .. N3-1 (taps at 6 e), then the m = 5 and m = 10 stages
    if constexpr (N3 >= 2) FE_STAGE(1, 3, 6);
    if constexpr (N3 >= 3) FE_STAGE(2, 3, 12);
    if constexpr (N3 >= 4) FE_STAGE(3, 3, 18);
    if constexpr (N3 >= 5) FE_STAGE(4, 3, 24);
    if constexpr (TAIL) { FE_STAGE(N3, 5, 6 * N3); FE_STAGE(N3 + 1, 10, 6 * N3 + 10); }
#undef FE_STAGE
    constexpr int NLAST = (N0 / 2) >> (H - 1);
    constexpr int PLAST = H == 1 ? 8 : (NLAST >= NT ? NLAST / NT : 1);
    constexpr int GS = PLAST >= 8 ? 3 : (PLAST == 4 ? 2 : 1);                   // final layout L(1 << GS)
    const cf *fin = ((H - 1) & 1) ? R1 : R0;                                      // stage e writes R1 when e is odd

    if constexpr (MODE == FE_L1) {
        // pairs of adjacent samples per lane through 16-byte stores (8-byte stores run at ~0.6x the rate); pairs start at
        // even ring positions, so they are aligned and never straddle the ring end
        cf *__restrict__ out = (cf *)p.out;
        const int nown = (int)((qa + p.TQ <= p.Q) ? p.TQ : (p.Q > qa ? p.Q - qa : 0));   // samples this tile stores
        const auto ld = [&](int i) { return fin[(p.HhQ + i) + ((p.HhQ + i) >> GS)]; };
        const int head = (int)((p.out_pos0 + qa) & 1ull) && nown > 0;
        if (head && tid == 0) out[(p.out_pos0 + qa) & p.out_mask] = ld(0);
        const int npair = (nown - head) >> 1;
        for (int t = tid; t < npair; t += NT) {
            const int i = head + 2 * t;
            const cf a = ld(i), b = ld(i + 1);
            *reinterpret_cast<float4 *>(out + ((p.out_pos0 + qa + i) & p.out_mask)) = make_float4(a.x, a.y, b.x, b.y);
        }
        if (((nown - head) & 1) && tid == 0) out[(p.out_pos0 + qa + nown - 1) & p.out_mask] = ld(nown - 1);
    } else {
        // ---- phase D: arbitrary resampler; a thread owns outputs ja + tid and ja + tid + 256 (taps already here) ----
        cf *__restrict__ out = (cf *)p.out;
        const auto resamp = [&](unsigned long long j, const float *bk) {
            const unsigned long long ph = (unsigned long long)p.phi0 + j * p.step;
            const int ql = (int)((ph >> 24) - qa) + p.HhQ - 13;
            cf y = cfm(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 14; k++) y = cfma(bk[k], fin[(ql + k) + ((ql + k) >> GS)], y);
            return y;
        };
        const unsigned long long j0 = pairs ? jh + 2ull * tid : ja + tid;
        if (pairs && j0 + 1 < jb) {
            const cf y0 = resamp(j0, bk0), y1 = resamp(j0 + 1, bk1);
            *reinterpret_cast<float4 *>(out + ((p.out_pos0 + j0) & p.out_mask)) = make_float4(y0.x, y0.y, y1.x, y1.y);
        } else if (j0 < jb) {
            out[(p.out_pos0 + j0) & p.out_mask] = resamp(j0, bk0);
        }
        // rare leftovers, taps fetched on the spot: the odd head sample, and anything beyond 2 NT outputs per tile (r_a > 1)
        if (tid == 0 && jh > ja && ja < jb) {
            const unsigned long long ph = (unsigned long long)p.phi0 + ja * p.step;
            out[(p.out_pos0 + ja) & p.out_mask] = resamp(ja, p.arb_bank + ((unsigned)(ph & 0xffffffu) >> 16) * 14u);
        }
        for (unsigned long long j = (pairs ? jh + 2ull * NT : ja + NT) + tid; j < jb; j += NT) {
            const unsigned long long ph = (unsigned long long)p.phi0 + j * p.step;
            out[(p.out_pos0 + j) & p.out_mask] = resamp(j, p.arb_bank + ((unsigned)(ph & 0xffffffu) >> 16) * 14u);
        }
    }
    if (MODE != FE_L2 && c == 0 && p.new_hist) {
        cf *__restrict__ nh = (cf *)p.new_hist;
        for (int i = tid; i < p.hcap; i += NT) {
            const long sb = (long)i + (long)p.n_in - (long)p.hcap;
            nh[i] = sb < 0 ? hist[(long)i + p.n_in] : x[sb];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tile carries from the probes.  W_c = v just before tile c's OWN range, V_c = v just before its halo:
//   W_{c+1} = rho W_c + (probeB_c - rho probeA_c),  rho = lambda^T_own;   V_c = (W_c - probeA_c) lambda^-Hh
// rho^k vanishes after K terms, so every tile sums its K predecessors independently (no serial chain).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fe_tiles(pmr_fe_tiles_params p)
{
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= p.ntiles) return;
    const cf *pa = (const cf *)p.probeA, *pb = (const cf *)p.probeB;
    const cf vs = *(const cf *)p.v_in, pl = *(const cf *)p.probeL;
    const float V0r = (vs.x - pl.x) * p.inv_lamL, V0i = (vs.y - pl.y) * p.inv_lamL;
    const cf a0 = pa[0];
    const float W0r = fmaf(p.lamHh, V0r, a0.x), W0i = fmaf(p.lamHh, V0i, a0.y);
    float ar = 0.f, ai = 0.f, pw = 1.f;
    const unsigned kmax = c < p.K ? c : p.K;
    for (unsigned k = 1; k <= kmax; k++) {
        const cf A = pa[c - k], B = pb[c - k];
        ar = fmaf(pw, fmaf(-p.rho, A.x, B.x), ar);
        ai = fmaf(pw, fmaf(-p.rho, A.y, B.y), ai);
        pw *= p.rho;
    }
    if (c <= p.K) { ar = fmaf(pw, W0r, ar); ai = fmaf(pw, W0i, ai); }
    const cf Ac = pa[c];
    const float Vr = (ar - Ac.x) * p.inv_lamHh, Vi = (ai - Ac.y) * p.inv_lamHh;
    ((cf *)p.V)[c] = cfm(Vr, Vi);
    if (c == p.c_end) {
        const cf pe = *(const cf *)p.probeE;
        *(cf *)p.v_out = cfm(fmaf(p.lamEnd, Vr, pe.x), fmaf(p.lamEnd, Vi, pe.y));
    }
}

// resampled-domain dc correction: xr[j] -= V_c * K * mu^q' * GA[idx_j]
__global__ __launch_bounds__(256) void k_fe_dcfix(pmr_fe_fix_params p)
{
    const unsigned j = p.j0 + blockIdx.x * 256u + threadIdx.x;
    if (j >= p.ny) return;
    // step != 0: j indexes resampler outputs (decimated index and polyphase gain from the 24-bit phase);
    // step == 0: j indexes decimated samples directly (the level-1 ring of a two-level front end)
    const unsigned long long ph = (unsigned long long)p.phi0 + (unsigned long long)j * p.step;
    const unsigned q = p.step ? (unsigned)(ph >> 24) : j;
    const unsigned idx = (unsigned)(ph & 0xffffffu) >> 16;
    const unsigned c = q / p.TQ;
    const unsigned ql = q - c * p.TQ + p.HhQ;
    const float g = p.Kgain * (p.step ? p.GA[idx] : 1.0f) * (p.T1[ql >> 5] * p.T2[ql & 31]);
    const cf V = ((const cf *)p.V)[c];
    cf *o = (cf *)p.xr + ((p.pos0 + j) & p.mask);
    cf v = *o;
    v.x = fmaf(-V.x, g, v.x);
    v.y = fmaf(-V.y, g, v.y);
    *o = v;
}

// ---------------------------------------------------------------------------------------------
// k_fe_tiles + k_fe_dcfix in one launch, one WAVE per front-end tile: the lanes sum the K predecessor terms of the
// tile's carry (wave reduction), then the same wave corrects the ~T_own * rate resampler outputs its tile produced, in
// place.  q' is simply the tile-local decimated index here, so the 32-bit division per sample of k_fe_dcfix is gone, and
// the consumer (channelizer) no longer has to apply the carry while staging.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fe_tilefix(pmr_fe_tiles_params t, pmr_fe_fix_params f, unsigned n_q /*decimated samples of the block*/,
                                                    int ablate /*timing experiments: 1 no carry sum, 2 no data pass*/)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c >= t.ntiles) return;
    const cf *pa = (const cf *)t.probeA, *pb = (const cf *)t.probeB;
    // ---- V_c exactly as k_fe_tiles computes it, the K-term sum spread over the lanes ----
    const unsigned kmax = (ablate & 1) ? 0u : (c < t.K ? c : t.K);
    float ar = 0.f, ai = 0.f;
    for (unsigned k = 1 + lane; k <= kmax; k += 64u) {
        const cf A = pa[c - k], B = pb[c - k];
        const float pw = t.rho_pow[k - 1];                           // rho^(k-1), tabulated in double on the host
        ar = fmaf(pw, fmaf(-t.rho, A.x, B.x), ar);
        ai = fmaf(pw, fmaf(-t.rho, A.y, B.y), ai);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ar += __shfl_xor(ar, d); ai += __shfl_xor(ai, d); }
    if (c <= t.K) {
        const cf vs = *(const cf *)t.v_in, pl = *(const cf *)t.probeL, a0 = pa[0];
        const float V0r = (vs.x - pl.x) * t.inv_lamL, V0i = (vs.y - pl.y) * t.inv_lamL;
        const float W0r = fmaf(t.lamHh, V0r, a0.x), W0i = fmaf(t.lamHh, V0i, a0.y);
        const float pw = t.rho_pow[kmax];
        ar = fmaf(pw, W0r, ar); ai = fmaf(pw, W0i, ai);
    }
    const cf Ac = pa[c];
    const float Vr = (ar - Ac.x) * t.inv_lamHh, Vi = (ai - Ac.y) * t.inv_lamHh;
    if (lane == 0) {
        ((cf *)t.V)[c] = cfm(Vr, Vi);
        if (c == t.c_end) {
            const cf pe = *(const cf *)t.probeE;
            *(cf *)t.v_out = cfm(fmaf(t.lamEnd, Vr, pe.x), fmaf(t.lamEnd, Vi, pe.y));
        }
    }
    // ---- the tile's own outputs [ja, jb): the range the front-end kernel published for this tile ----
    if (ablate & 2) return;
    const unsigned long long qa = (unsigned long long)c * f.TQ;
    if (qa >= n_q) return;
    const unsigned long long ja = ((const unsigned long long *)t.tile_j)[2 * c], jb = ((const unsigned long long *)t.tile_j)[2 * c + 1];
    cf *xr = (cf *)f.xr;
    // A lane corrects PAIRS of adjacent outputs with one 16-byte load and store (8-byte accesses run at ~0.6x the rate of
    // 16-byte ones); a pair never straddles the ring end because the ring size is even and pairs start at even ring
    // positions.  Three pairs per lane per batch, all loads of a batch in flight before the first is consumed.
    const auto gain = [&](unsigned long long j) {
        const unsigned long long ph = (unsigned long long)f.phi0 + j * f.step;
        const unsigned ql = (unsigned)((ph >> 24) - qa) + f.HhQ;
        return f.GA[(unsigned)(ph & 0xffffffu) >> 16] * (f.T1[ql >> 5] * f.T2[ql & 31]);
    };
    unsigned long long js = ja;
    if (js < jb && ((f.pos0 + js) & 1ull)) {                    // odd ring position: one single sample first
        if (lane == 0) {
            cf *o = xr + ((f.pos0 + js) & f.mask);
            const float gg = f.Kgain * gain(js);
            cf w = *o;
            w.x = fmaf(-Vr, gg, w.x); w.y = fmaf(-Vi, gg, w.y);
            *o = w;
        }
        js++;
    }
    const unsigned long long npair = (jb - js) >> 1;
    for (unsigned long long pb = lane; pb < npair; pb += 192u) {
        float4 v[3]; float g0[3], g1[3]; float4 *o[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const unsigned long long pi = pb + 64u * u;
            const unsigned long long j = js + 2ull * (pi < npair ? pi : 0ull);
            o[u] = reinterpret_cast<float4 *>(xr + ((f.pos0 + j) & f.mask));
            v[u] = *o[u];
            g0[u] = gain(j); g1[u] = gain(j + 1);
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if (pb + 64u * u < npair) {
                const float ga = f.Kgain * g0[u], gb = f.Kgain * g1[u];
                float4 w = v[u];
                w.x = fmaf(-Vr, ga, w.x); w.y = fmaf(-Vi, ga, w.y);
                w.z = fmaf(-Vr, gb, w.z); w.w = fmaf(-Vi, gb, w.w);
                *o[u] = w;
            }
        }
    }
    if (((jb - js) & 1ull) && lane == 0) {                      // odd count: the last sample alone
        const unsigned long long j = jb - 1;
        cf *o = xr + ((f.pos0 + j) & f.mask);
        const float gg = f.Kgain * gain(j);
        cf w = *o;
        w.x = fmaf(-Vr, gg, w.x); w.y = fmaf(-Vi, gg, w.y);
        *o = w;
    }
}

// raw history for the next call: last hcap samples of (old history || block)
__global__ __launch_bounds__(256) void k_fe_hist(const cf *__restrict__ old_hist, const cf *__restrict__ x,
                                                 unsigned n_in, cf *__restrict__ new_hist, unsigned hcap)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= hcap) return;
    const long s = (long)i + (long)n_in - (long)hcap;        // block-relative index
    new_hist[i] = s < 0 ? old_hist[(long)i + n_in] : x[s];
}

// ---------------------------------------------------------------------------------------------
template <int NT, int SPT, int MODE>
static int launch_frontend_t(hipStream_t st, const pmr_fe_params *p, unsigned ntiles)
{
    const size_t n0 = (size_t)NT * SPT;
    size_t lds = (FE_PAD + n0 + n0 / SPT + 32 + 10 * (NT / 64 + 1)) * sizeof(cf);   /* pad + tile + scan scratch + halo exchange */
    { static long extra = -1; if (extra < 0) { const char *e = getenv("PMR_FE_LDS_EXTRA"); extra = e ? atol(e) : 0; } lds += (size_t)extra; }   /* experiment: fewer tiles per CU */
    static unsigned long long attr_set = 0;
    if (pmr_attr_needed(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_frontend<NT, SPT, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    auto kern = k_frontend<NT, SPT, MODE>;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), lds, st, *p);
    return (int)hipGetLastError();
}

/* tile geometries: (threads, samples per thread).  4096-sample tiles as 512 x 8 (default: twice the waves per CU
 * of 256 x 16 for the same LDS) or 256 x 16; 16384-sample tiles (deep cascades) as 1024 x 16.                 */
template <int MODE, int N3, int TAIL>
static int launch_frontend_fast(hipStream_t st, const pmr_fe_params *p, unsigned ntiles)
{
    const size_t n0 = 4096;
    const size_t body = MODE == FE_L2 ? n0 + n0 / 16 : (n0 / 2 + n0 / 16) + (n0 / 4 + n0 / 16);   /* whole raw tile | R0 + R1 */
    size_t lds = (FE_PAD + body + 11 * (256 / 64)) * sizeof(cf);
    static unsigned long long attr_set = 0;
    if (pmr_attr_needed(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_frontend_fast<MODE, N3, TAIL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    auto kern = k_frontend_fast<MODE, N3, TAIL>;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, st, *p);
    return (int)hipGetLastError();
}

/* the specialised kernel covers: N3 six-tap stages, then optionally (m = 5, m = 10); 256 x 16 tiles */
static int fast_pattern(const pmr_fe_params *p, int *n3, int *tail)
{
    int k = 0;
    while (k < p->h && p->m[k] == 3) k++;
    *n3 = k;
    if (k == p->h) { *tail = 0; return k >= 1; }
    if (k >= 1 && k + 2 == p->h && p->m[k] == 5 && p->m[k + 1] == 10) { *tail = 1; return 1; }
    return 0;
}

extern "C" int pmr_launch_frontend(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, int nt, int spt)
{
    if (!ntiles) return 0;
    hipStream_t st = (hipStream_t)s;
    static int use_fast = -1;
    if (use_fast < 0) { const char *e = getenv("PMR_FE_KERNEL"); use_fast = !(e && !strcmp(e, "generic")); }
    int n3 = 0, tail = 0;
    if (use_fast && nt == 256 && spt == 16 && !p->ablate && !p->stamps && p->taps_valid && fast_pattern(p, &n3, &tail)) {
        if (p->mode == FE_FULL && tail) {
            if (n3 == 1) return launch_frontend_fast<FE_FULL, 1, 1>(st, p, ntiles);
            if (n3 == 2) return launch_frontend_fast<FE_FULL, 2, 1>(st, p, ntiles);
            if (n3 == 3) return launch_frontend_fast<FE_FULL, 3, 1>(st, p, ntiles);
        }
        if (p->mode == FE_L2 && tail && n3 == 1) return launch_frontend_fast<FE_L2, 1, 1>(st, p, ntiles);
        if (p->mode == FE_L1 && !tail) {
            if (n3 == 2) return launch_frontend_fast<FE_L1, 2, 0>(st, p, ntiles);
            if (n3 == 3) return launch_frontend_fast<FE_L1, 3, 0>(st, p, ntiles);
            if (n3 == 4) return launch_frontend_fast<FE_L1, 4, 0>(st, p, ntiles);
        }
    }
    if (p->mode == FE_L1 && nt == 256 && spt == 16) return launch_frontend_t<256, 16, FE_L1>(st, p, ntiles);
    if (p->mode == FE_L2 && nt == 256 && spt == 16) return launch_frontend_t<256, 16, FE_L2>(st, p, ntiles);
    if (p->mode != FE_FULL) return (int)hipErrorInvalidValue;
    if (nt == 256 && spt == 16) return launch_frontend_t<256, 16, FE_FULL>(st, p, ntiles);
    if (nt == 1024 && spt == 16) return launch_frontend_t<1024, 16, FE_FULL>(st, p, ntiles);
    if (nt == 512 && spt == 16) return launch_frontend_t<512, 16, FE_FULL>(st, p, ntiles);
    if (nt == 512 && spt == 8) return launch_frontend_t<512, 8, FE_FULL>(st, p, ntiles);
    if (nt == 256 && spt == 8) return launch_frontend_t<256, 8, FE_FULL>(st, p, ntiles);
    if (nt == 128 && spt == 16) return launch_frontend_t<128, 16, FE_FULL>(st, p, ntiles);
    if (nt == 192 && spt == 16) return launch_frontend_t<192, 16, FE_FULL>(st, p, ntiles);
    return (int)hipErrorInvalidValue;
}

extern "C" int pmr_launch_fe_tiles(pmr_stream_t s, const pmr_fe_tiles_params *p)
{
    if (!p->ntiles) return 0;
    hipLaunchKernelGGL(k_fe_tiles, dim3((p->ntiles + 255) / 256), dim3(256), 0, (hipStream_t)s, *p);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_fe_dcfix(pmr_stream_t s, const pmr_fe_fix_params *p)
{
    if (p->ny <= p->j0) return 0;
    hipLaunchKernelGGL(k_fe_dcfix, dim3((p->ny - p->j0 + 255) / 256), dim3(256), 0, (hipStream_t)s, *p);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_fe_tilefix(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f, unsigned n_q)
{
    if (!t->ntiles) return 0;
    const char *e = getenv("PMR_TILEFIX_ABLATE");
    hipLaunchKernelGGL(k_fe_tilefix, dim3((t->ntiles + 3) / 4), dim3(256), 0, (hipStream_t)s, *t, *f, n_q, e ? atoi(e) : 0);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_fe_hist(pmr_stream_t s, const void *old_hist, const void *x, unsigned n_in,
                                  void *new_hist, unsigned hcap)
{
    hipLaunchKernelGGL(k_fe_hist, dim3((hcap + 255) / 256), dim3(256), 0, (hipStream_t)s, (const cf *)old_hist,
                       (const cf *)x, n_in, (cf *)new_hist, hcap);
    return (int)hipGetLastError();
}
