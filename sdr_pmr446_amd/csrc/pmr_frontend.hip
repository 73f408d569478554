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
#include <stdlib.h>
#include <string.h>

#include "pmr_fe_common.hpp"

// ---------------------------------------------------------------------------------------------
// one half-band stage out of LDS, P outputs per thread, 2*MM branch taps   
//   z1[o] = z0[2o+1-2m] + sum_j h1[j] * z0[2o - 2(2m-1-j)]
// reads layout L(2P), writes layout L(max(P,2)) in place (two barriers).
// ---------------------------------------------------------------------------------------------
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
// NT threads own a tile of N0 = NT * SPT input samples (SPT consecutive samples per thread in the dc scan).
// MODE 0: whole front end (raw in -> dc-block -> all stages -> resampler).
// Deep cascades (tile halo would eat the tile) run as two launches of the same code:
// MODE 1: level 1 = raw in -> dc-block -> the first few (6-tap) stages -> decimated stream stored to a ring;
// MODE 2: level 2 = ring in (dc carry of level 1 applied at load) -> remaining stages -> resampler.
enum { FE_FULL = 0, FE_L1 = 1, FE_L2 = 2 };

template <int NT, int SPT, int MODE>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 1) void k_frontend(pmr_fe_params p)
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

    // ---- resampler bookkeeping first: which outputs this tile owns, and the polyphase taps of this thread's first
    // output (a dependent global load: issued now, it lands while the raw tile streams in) ----
    const unsigned long long qa = (unsigned long long)c * p.TQ;
    unsigned long long ja = 0, jb = 0;
    {
        unsigned long long qb = qa + p.TQ;
        if (qb > p.Q) qb = p.Q;
        if (MODE != FE_L1 && qa < qb) {
            const unsigned long long sa = qa << 24, sb = qb << 24;
            ja = sa <= p.phi0 ? 0ull : ceil_div_step(sa - p.phi0, p.step, p.step_rinv);
            jb = sb <= p.phi0 ? 0ull : ceil_div_step(sb - p.phi0, p.step, p.step_rinv);
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
                if (V1 && jn >= 0 && jn < (long long)p.fix_limit) {
                    const unsigned c1 = (unsigned)jn / p.fix_TQ;
                    const unsigned ql = (unsigned)jn - c1 * p.fix_TQ + p.fix_HhQ;
                    const float g = p.fix_K * (p.fix_T1[ql >> 5] * p.fix_T2[ql & 31]);
                    const cf Vc = V1[c1];
                    v = cf{fmaf(-Vc.x, g, v.x), fmaf(-Vc.y, g, v.y)};
                }
            }
            buf[lidx<SPT>(i)] = v;
        }
    } else {
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

    // ---- phase B: dc blocker (:795) from zero state.  With a 6-tap first stage (m = 3, every cascade of >= 3 stages)
    // the first half-band stage is computed right here from registers: no write-back of yb, no window re-read. ----
    cf *bnd = wagg + NT / 64;                                // [NT/64][10] last 10 dc-blocked samples of every wave
    const bool fuse0 = SPT == 16 && p.h >= 1 && p.m[0] == 3;
    {
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

    // ---- phase C: half-band cascade, in place (stage e halves the sample count) ----
    int g_shift = SPT == 16 ? 4 : 3;                         // layout of the current signal: L(1 << g_shift)
    {
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
    if (MODE != FE_L1) {
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
}

// ---------------------------------------------------------------------------------------------
// Tile carries from the probes.  W_c = v just before tile c's OWN range, V_c = v just before its halo:
//   W_{c+1} = rho W_c + (probeB_c - rho probeA_c),  rho = lambda^T_own;   V_c = (W_c - probeA_c) lambda^-Hh
// rho^k vanishes after K terms (K tabulated on the host: rho^K < 1e-12), so every tile sums its K predecessors independently
// -- no serial chain.  One thread evaluates one carry.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ cf fe_carry_V(const pmr_fe_tiles_params &p, unsigned c)
{
    const cf *pa = (const cf *)p.probeA, *pb = (const cf *)p.probeB;
    float ar = 0.f, ai = 0.f;
    const unsigned kmax = c < p.K ? c : p.K;
    if (p.K <= 16) {
        // every probe of the look-back is requested before the first is used: the loop below with a run-time bound would pay the
        // L2 latency of its loads one after the other (15 in a row for a 6 us kernel)
        cf A[16], B[16]; float pw[16];
#pragma unroll
        for (unsigned k = 1; k <= 16; k++) {
            const unsigned i = k <= kmax ? c - k : c;
            A[k - 1] = pa[i]; B[k - 1] = pb[i]; pw[k - 1] = k <= kmax ? p.rho_pow[k - 1] : 0.f;
        }
#pragma unroll
        for (unsigned k = 1; k <= 16; k++) {
            ar = fmaf(pw[k - 1], fmaf(-p.rho, A[k - 1].x, B[k - 1].x), ar);
            ai = fmaf(pw[k - 1], fmaf(-p.rho, A[k - 1].y, B[k - 1].y), ai);
        }
    } else {
        for (unsigned k = 1; k <= kmax; k++) {
            const cf A = pa[c - k], B = pb[c - k];
            const float pw = p.rho_pow[k - 1];                           // rho^(k-1), tabulated in double on the host
            ar = fmaf(pw, fmaf(-p.rho, A.x, B.x), ar);
            ai = fmaf(pw, fmaf(-p.rho, A.y, B.y), ai);
        }
    }
    if (c <= p.K) {
        const cf vs = *(const cf *)p.v_in, pl = *(const cf *)p.probeL, a0 = pa[0];
        const float V0r = (vs.x - pl.x) * p.inv_lamL, V0i = (vs.y - pl.y) * p.inv_lamL;
        const float W0r = fmaf(p.lamHh, V0r, a0.x), W0i = fmaf(p.lamHh, V0i, a0.y);
        const float pw = p.rho_pow[kmax];
        ar = fmaf(pw, W0r, ar); ai = fmaf(pw, W0i, ai);
    }
    const cf Ac = pa[c];
    return cfm((ar - Ac.x) * p.inv_lamHh, (ai - Ac.y) * p.inv_lamHh);
}

// Level-1 carries of the two-level front end, and the dc state handed to the next call.  The same launch corrects, IN PLACE,
// the last few new samples of the level-1 ring -- what the next call's level 2 re-reads as history (this call's level 2 applies
// the carry itself while loading and skips those samples: pmr_fe_params.fix_limit):  ring[j] -= V_c * K1 * mu^q',  c = j / TQ.
// The handful of tail threads recompute the carries they need, so there is no dependency between workgroups.
__global__ __launch_bounds__(256) void k_fe_carry(pmr_fe_tiles_params t, pmr_fe_fix_params f, unsigned nb_tiles)
{
    if (blockIdx.x < nb_tiles) {
        const unsigned c = blockIdx.x * 256u + threadIdx.x;
        if (c >= t.ntiles) return;
        const cf V = fe_carry_V(t, c);
        ((cf *)t.V)[c] = V;
        if (c == t.c_end) {
            const cf pe = *(const cf *)t.probeE;
            *(cf *)t.v_out = cfm(fmaf(t.lamEnd, V.x, pe.x), fmaf(t.lamEnd, V.y, pe.y));
        }
        return;
    }
    const unsigned j = f.j0 + (blockIdx.x - nb_tiles) * 256u + threadIdx.x;
    if (j >= f.ny) return;
    const unsigned c = j / f.TQ, ql = j - c * f.TQ + f.HhQ;
    const float g = f.Kgain * (f.T1[ql >> 5] * f.T2[ql & 31]);
    const cf V = fe_carry_V(t, c);
    cf *o = (cf *)f.xr + ((f.pos0 + j) & f.mask);
    cf v = *o;
    v.x = fmaf(-V.x, g, v.x);
    v.y = fmaf(-V.y, g, v.y);
    *o = v;
}

// One-level front end, carry applied where the channelizer loads the resampled stream (pmr_carry_fix): this launch computes the
// tile carries V_c (one thread each) + the dc state for the next call, and corrects IN PLACE only the block's last outputs
// [j0, ny) -- what later calls re-read as filter history -- one thread per output, in k_fe_tilefix's arithmetic.
__global__ __launch_bounds__(256) void k_fe_carry_tail(pmr_fe_tiles_params t, pmr_fe_fix_params f, unsigned nb_tiles)
{
    if (blockIdx.x < nb_tiles) {
        const unsigned c = blockIdx.x * 256u + threadIdx.x;
        if (c >= t.ntiles) return;
        const cf V = fe_carry_V(t, c);
        ((cf *)t.V)[c] = V;
        if (c == t.c_end) {
            const cf pe = *(const cf *)t.probeE;
            *(cf *)t.v_out = cfm(fmaf(t.lamEnd, V.x, pe.x), fmaf(t.lamEnd, V.y, pe.y));
        }
        return;
    }
    const unsigned j = f.j0 + (blockIdx.x - nb_tiles) * 256u + threadIdx.x;
    if (j >= f.ny) return;
    const unsigned long long ph = (unsigned long long)f.phi0 + (unsigned long long)j * f.step;
    const unsigned qd = (unsigned)(ph >> 24), c = qd / f.TQ, ql = qd - c * f.TQ + f.HhQ;
    const float gg = f.GA[(unsigned)(ph & 0xffffffu) >> 16] * (f.T1[ql >> 5] * f.T2[ql & 31]);    // GA carries the cascade gain
    const cf V = fe_carry_V(t, c);
    cf *o = (cf *)f.xr + ((f.pos0 + j) & f.mask);
    cf w = *o;
    w.x = fmaf(-V.x, gg, w.x); w.y = fmaf(-V.y, gg, w.y);
    *o = w;
}

// ---------------------------------------------------------------------------------------------
// One-level front end: carries + correction of the resampled stream in one launch, one WAVE per front-end tile.  The lanes sum
// the K predecessor terms of the tile's carry (wave reduction), then the same wave corrects the ~T_own * rate resampler outputs
// its tile produced, in place:  xr[j] -= V_c * K * mu^q' * GA[idx_j]   (q' = tile-local decimated index, idx_j = polyphase
// branch of output j; K, mu, GA: closed-form gains of the cascade for the exponential the missing carry adds).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fe_tilefix(pmr_fe_tiles_params t, pmr_fe_fix_params f, unsigned n_q /*decimated samples of the block*/)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c >= t.ntiles) return;
    // ---- V_c: fe_carry_V's arithmetic (every lane evaluates the same sum), so that this pass and k_fe_carry_tail -- the form
    // that leaves most of the block to the channelizer's loads -- correct a sample with the same carry, bit for bit ----
    const cf Vc = fe_carry_V(t, c);
    const float Vr = Vc.x, Vi = Vc.y;
    if (lane == 0) {
        ((cf *)t.V)[c] = cfm(Vr, Vi);
        if (c == t.c_end) {
            const cf pe = *(const cf *)t.probeE;
            *(cf *)t.v_out = cfm(fmaf(t.lamEnd, Vr, pe.x), fmaf(t.lamEnd, Vi, pe.y));
        }
    }
    // ---- the tile's own outputs [ja, jb): the range the front-end kernel published for this tile ----
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
    unsigned long long js = ja > f.j0 ? ja : f.j0;             // carry applied at load (pmr_carry_fix): only the block's tail is fixed here
    if (js >= jb) return;
    if ((f.pos0 + js) & 1ull) {                    // odd ring position: one single sample first
        if (lane == 0) {
            cf *o = xr + ((f.pos0 + js) & f.mask);
            const float gg = gain(js);
            cf w = *o;
            w.x = fmaf(-Vr, gg, w.x); w.y = fmaf(-Vi, gg, w.y);
            *o = w;
        }
        js++;
    }
    const unsigned long long npair = (jb - js) >> 1;
    for (unsigned long long pb2 = lane; pb2 < npair; pb2 += 192u) {
        float4 v[3]; float g0[3], g1[3]; float4 *o[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const unsigned long long pi = pb2 + 64u * u;
            const unsigned long long j = js + 2ull * (pi < npair ? pi : 0ull);
            o[u] = reinterpret_cast<float4 *>(xr + ((f.pos0 + j) & f.mask));
            v[u] = *o[u];
            g0[u] = gain(j); g1[u] = gain(j + 1);
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            if (pb2 + 64u * u < npair) {
                const float ga = g0[u], gb = g1[u];
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
        const float gg = gain(j);
        cf w = *o;
        w.x = fmaf(-Vr, gg, w.x); w.y = fmaf(-Vi, gg, w.y);
        *o = w;
    }
}

// ---------------------------------------------------------------------------------------------
template <int NT, int SPT, int MODE>
static int launch_frontend_t(hipStream_t st, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev = nullptr)
{
    const size_t n0 = (size_t)NT * SPT;
    const size_t lds = (FE_PAD + n0 + n0 / SPT + 32 + 10 * (NT / 64 + 1)) * sizeof(cf);   /* pad + tile + scan scratch + halo exchange */
    static pmr_attr_flags attr_set{0};
    if (pmr_attr_needed(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_frontend<NT, SPT, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    auto kern = k_frontend<NT, SPT, MODE>;
    PMR_LAUNCH_EV(kern, dim3(ntiles), dim3(NT), lds, st, ev, *p);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_fe_level2_fast(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles);   /* pmr_fe_fast.hip */

/* tile geometries: (threads, samples per thread).  4096-sample tiles as 256 x 16; 16384-sample tiles (cascades too deep
 * for those where the two-level split does not apply) as 1024 x 16.  The specialised kernels of pmr_fe_fast.hip take the
 * cascades they cover unless `generic` is set (PMR_FE_KERNEL=generic).                                               */
extern "C" int pmr_launch_frontend(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, int nt, int spt,
                                   const pmr_launch_events *ev)
{
    if (!ntiles) return 0;
    hipStream_t st = (hipStream_t)s;
    if (nt == 256 && spt == 16 && (p->mode == FE_FULL || p->mode == FE_L1)) {
        const int rc = pmr_launch_fe_fast(s, p, ntiles, ev);        /* the specialised kernels, where the cascade is one they cover */
        if (rc >= 0) return rc;
    }
    if (p->in_fmt) return (int)hipErrorInvalidValue;                /* (only k_fe_fast converts integer samples as it loads) */
    if (p->mode == FE_L1 && nt == 256 && spt == 16) return launch_frontend_t<256, 16, FE_L1>(st, p, ntiles, ev);
    if (p->mode == FE_L2 && nt == 256 && spt == 16) return launch_frontend_t<256, 16, FE_L2>(st, p, ntiles, ev);
    if (p->mode != FE_FULL) return (int)hipErrorInvalidValue;
    if (nt == 256 && spt == 16) return launch_frontend_t<256, 16, FE_FULL>(st, p, ntiles, ev);
    if (nt == 1024 && spt == 16) return launch_frontend_t<1024, 16, FE_FULL>(st, p, ntiles, ev);
    return (int)hipErrorInvalidValue;
}

extern "C" int pmr_launch_frontend_l2(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, int fast)
{
    if (!ntiles) return 0;
    if (fast) return pmr_launch_fe_level2_fast(s, p, ntiles);
    return launch_frontend_t<256, 16, FE_L2>((hipStream_t)s, p, ntiles);
}

extern "C" int pmr_launch_fe_carry(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f)
{
    if (!t->ntiles) return 0;
    const unsigned nb_tiles = (t->ntiles + 255) / 256, nb_fix = f->ny > f->j0 ? (f->ny - f->j0 + 255) / 256 : 0;
    PMR_KLAUNCH(k_fe_carry, dim3(nb_tiles + nb_fix), dim3(256), 0, (hipStream_t)s, *t, *f, nb_tiles);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_fe_carry_tail(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f,
                                        const pmr_launch_events *ev)
{
    if (!t->ntiles) return 0;
    const unsigned nb_tiles = (t->ntiles + 255) / 256, nb_fix = f->ny > f->j0 ? (f->ny - f->j0 + 255) / 256 : 0;
    PMR_LAUNCH_EV(k_fe_carry_tail, dim3(nb_tiles + nb_fix), dim3(256), 0, (hipStream_t)s, ev, *t, *f, nb_tiles);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_fe_tilefix(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f, unsigned n_q,
                                     const pmr_launch_events *ev)
{
    if (!t->ntiles) return 0;
    PMR_LAUNCH_EV(k_fe_tilefix, dim3((t->ntiles + 3) / 4), dim3(256), 0, (hipStream_t)s, ev, *t, *f, n_q);
    return (int)hipGetLastError();
}
