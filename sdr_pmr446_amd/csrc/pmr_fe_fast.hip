// pmr_fe_fast.hip -- the SPECIALISED front-end kernels for gfx950: dc-block -> half-band cascade -> arbitrary resampler
// (reference src/sdr_pmr446.c:795-796: iirfilt_crcf_execute_block + msresamp_crcf_execute) for the cascades liquid's
// As = 60 dB design produces: N3 six-tap stages (m = 3), then the m = 5 and m = 10 stages, then the 256 x 14 polyphase bank.
//
//   k_fe_fast<FE_FULL, N3, 1>   whole front end in one pass over the raw block (cfg2: N3 = 1, cfg3: N3 = 2; N3 = 0: the reference's
//                               own 1.024 MS/s plan, whose cascade is just m = 5, m = 10)
//   k_fe_fast<FE_L1,  N3, 0>    level 1 of a deep cascade (cfg5 / dsd_in: N3 = 4): dc-block + N3 six-tap stages -> decimated ring
//   k_fe_level2                 level 2: ring in (level 1's dc carry applied at load) -> m = 5 -> m = 10 -> resampler
//
// Same arithmetic and the same order of operations as the run-time-parameterised k_frontend (pmr_frontend.hip); what is
// specialised is WHEN things are fetched and how much instruction overhead surrounds the ~13 packed MACs per sample:
//   * the raw tile goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass).  The DMA's
//     LDS image is lane-linear, so the bank-conflict-free layout is obtained by permuting the SOURCE addresses: 16-byte chunk
//     g of the tile lands in slot (g & ~7) | ((g & 7) ^ ((g >> 4) & 7)), and thread t reads its eight chunks 8t..8t+7 with
//     ds_read_b128 from slots 8t + (j ^ ((t >> 1) & 7)) -- 16 consecutive lanes then touch 16 distinct 16-byte bank groups.
//     Each 128-byte line is still fetched by 8 adjacent lanes of one instruction (the permutation stays inside a line);
//   * stage count, per-stage outputs per thread and LDS layouts are compile-time: every LDS address is thread base + immediate,
//     branch taps arrive as scalars from the kernel-argument segment;
//   * tile bookkeeping (which resampler outputs a tile owns) is wave-uniform INTEGER arithmetic on the scalar unit
//     (ceil_div_step) instead of an fp64 division in every lane;
//   * the cascade ping-pongs between two LDS regions (one barrier per stage); z1 / z2 reuse the raw tile's 32 KB.
#include <stdlib.h>
#include <string.h>

#include "pmr_fe_common.hpp"

enum { FE_FULL = 0, FE_L1 = 1 };
#ifndef FE_DMA_AUX
#define FE_DMA_AUX 2        /* cache policy of the raw-tile LDS-DMA: 2 = nt (streaming, read once), 0 = default.  The raw block is read
                               exactly once: with nt it no longer displaces the back-end kernels' working sets (ring, scratch,
                               discriminator rows) from L2 / Infinity Cache -- neutral for this kernel alone, +3.5 % for the pipelined
                               chain (cfg5 486 -> 503, cfg2 313 -> 325 GS/s on one box; sc0 / sc1 on top change nothing; non-temporal loads of
                               level 2's ring reads or non-temporal PCM stores: neutral to slightly worse, not used) */
#endif

#ifndef FE_LOOP_MINB
#define FE_LOOP_MINB 4      /* k_fe_loop's register cap: 4 -> <= 128 VGPRs (it takes 102 / 65), 5 -> <= 96 (two values spilled) */
#endif

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// slot of 16-byte chunk g (and chunk of slot g: the map is an involution)
static __device__ __forceinline__ int fe_swz(int g) { return (g & ~7) | ((g & 7) ^ ((g >> 4) & 7)); }

#ifdef FE_STAMP
// diagnostic build (tools/fe_phase_times.py): s_memtime at the phase boundaries of every tile, thread 0 -> fe_stamps[tile][8]
__device__ unsigned long long fe_stamps[65536 * 8];
#define FE_STAMP_AT(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) fe_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int pmr_debug_fe_stamps(void *dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fe_stamps), bytes, 0, hipMemcpyDeviceToHost); }
#else
#if defined(FE_STOP)        /* timing experiment (tools/variant_kstats.sh): the kernel up to phase boundary FE_STOP; WRONG results */
#define FE_STAMP_AT(i) do { if ((i) == FE_STOP && p.n_in != 0xffffffffu) return; } while (0)
#else
#define FE_STAMP_AT(i) do { } while (0)
#endif
#endif

// sample b of the caller's block in its own format (pmr_fe_params.in_fmt): the conversions of k_iq_convert (pmr_kernels.hip) /
// pmr_io.c, operation for operation, so a block converted here equals the block converted first and fed as cf32
static __device__ __forceinline__ cf fe_raw(const void *x, long b, int fmt)
{
    if (fmt == 0) return reinterpret_cast<const cf *>(x)[b];
    if (fmt == 1) {
        const unsigned w = reinterpret_cast<const unsigned *>(x)[b];
        return cfm((float)(short)(w & 0xffffu) * (1.0f / 32768.0f), (float)(short)(w >> 16) * (1.0f / 32768.0f));
    }
    const unsigned w = reinterpret_cast<const unsigned short *>(x)[b];
    return cfm(((float)(w & 0xffu) - 127.5f) * (1.0f / 127.5f), ((float)(w >> 8) - 127.5f) * (1.0f / 127.5f));
}

template <int MODE, int N3, int TAIL>
__global__ __launch_bounds__(256, 4) void k_fe_fast(pmr_fe_params p)
{
    static_assert(N3 >= 1 || (MODE == FE_FULL && TAIL == 1), "a cascade without six-tap stages is (m = 5, m = 10)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256, SPT = 16, N0 = NT * SPT;
    constexpr int H = N3 + 2 * TAIL;
    constexpr int R1_OFF = (N0 / 2) + (N0 / 2) / 8;         // z1 (2048 samples, layout L(8)) fills [0, R1_OFF) of the tile area
    // LDS: [FE_PAD zero pad | raw tile, 4096 samples = 2048 swizzled 16-byte chunks; afterwards R0 (z1, 2304 slots) and
    // R1 (z2, 1280 slots) | scan scratch] = 33.6 KB -> four tiles per CU
    // N3 == 0 (no six-tap stage: the reference's own 1.024 MS/s plan is m = 5, 10): the dc-blocked tile goes back to LDS in
    // layout L(16) (4352 slots) and the m = 5 stage runs from there in place; the scan scratch sits behind it
    constexpr int SCR = N3 == 0 ? N0 + N0 / 16 : N0;
    cf *buf = reinterpret_cast<cf *>(smem) + FE_PAD;
    cf *wagg = buf + SCR;
    cf *bnd = wagg + NT / 64;                                 // [4][10]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FE_STAMP_AT(0);
    const cf *__restrict__ x = (const cf *)p.x;              // (cf32 view: only used when p.in_fmt == 0)
    const cf *__restrict__ hist = (const cf *)p.hist;
    const float lam = -p.dc_a1;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so give every XCD a
    // CONTIGUOUS range of tiles -- a tile's halo is its left neighbour's tail and can then be an L2 hit instead of a second
    // HBM read (PMC: 8 % extra fetch without it).  Placement only affects speed, never results.
    int c = blockIdx.x;
    {
        const int nt_all = gridDim.x, per = nt_all >> 3, main = per << 3;
        if (c < main) c = (c & 7) * per + (c >> 3);
    }
#include "pmr_fe_tile.inc"
}

// ---------------------------------------------------------------------------------------------
// k_fe_loop: the same tile (pmr_fe_tile.inc), several tiles per workgroup (tiles_per_wg: the grid is ntiles / tiles_per_wg workgroups).  Why: a tile's last act is a global store,
// and a wave cannot retire before its stores are acknowledged -- behind the queue of tile DMAs that saturates the memory path that
// takes about a microsecond, during which the tile's 33.6 KB of LDS and four wave slots wait for nothing.  Measured (round 4,
// profiles/r04_ab_log.txt r4u): the level-1 kernel of cfg5 takes 96 us, 78 us with the ONE store instruction of a tile predicated
// off (and 78 us with all arithmetic removed as well: that is its DMA rate); cfg2's kernel 119 -> 93-99 us.  In a loop the next
// tile's DMA is requested right behind the stores and both complete under one wait; workgroup launch and retirement are paid
// once per several tiles.  Same statements as k_fe_fast for every sample: bit-identical (tests/test_gpu_fe_loop.py).
//   * the tile's parameters are read from the kernel-argument segment again in every trip (the empty asm hides the pointer's
//     provenance): hoisted out of the loop, the ~100 scalars a tile uses (taps, powers of lambda, geometry) do not fit the scalar
//     file and end up spilled to vector lanes -- round 2's persistent kernel carried 65 of those.
// ---------------------------------------------------------------------------------------------
template <int MODE, int N3, int TAIL>
__global__ __launch_bounds__(256, FE_LOOP_MINB) void k_fe_loop(const pmr_fe_params p_arg, unsigned ntiles)
{
    static_assert(N3 >= 1 || (MODE == FE_FULL && TAIL == 1), "a cascade without six-tap stages is (m = 5, m = 10)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256, SPT = 16, N0 = NT * SPT;
    constexpr int H = N3 + 2 * TAIL;
    constexpr int R1_OFF = (N0 / 2) + (N0 / 2) / 8;
    constexpr int SCR = N3 == 0 ? N0 + N0 / 16 : N0;           // (LDS layout: k_fe_fast)
    cf *buf = reinterpret_cast<cf *>(smem) + FE_PAD;
    cf *wagg = buf + SCR;
    cf *bnd = wagg + NT / 64;                                 // [4][10]

    // trip i of workgroup w takes tile i * gridDim + w': the resident workgroups walk CONSECUTIVE tiles at any moment, like k_fe_fast's
    // dispatch order (runs of tpw consecutive tiles per workgroup put the concurrent DMAs tpw * 31.5 KB apart -- the same few HBM
    // channels: 90 -> 97 -> 130 us for tpw = 1, 4, 16 at cfg5); w' = the XCD-contiguous renumbering inside a row
    int wg = blockIdx.x;
    {
        const int per = gridDim.x >> 3, main = per << 3;
        if (wg < main) wg = (wg & 7) * per + (wg >> 3);
    }
    const int c_first = wg, c_step = (int)gridDim.x;
#pragma nounroll
    for (int c = c_first; c < (int)ntiles; c += c_step) {
        const __attribute__((address_space(4))) pmr_fe_params *pk =
            (const __attribute__((address_space(4))) pmr_fe_params *)__builtin_amdgcn_kernarg_segment_ptr();   // (p_arg is the first argument)
        asm volatile("" : "+s"(pk));
        const pmr_fe_params &p = *(const pmr_fe_params *)pk;
        // ... and so is everything derived from the thread index (LDS addresses of every phase, swizzles, DMA offsets): kept live
        // across the loop they cost ~50 VGPRs, and the register count decides what shares a SIMD with these waves
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63, wave = tid >> 6;
        const cf *__restrict__ x = (const cf *)p.x;
        const cf *__restrict__ hist = (const cf *)p.hist;
        const float lam = -p.dc_a1;
        if (c != c_first) __syncthreads();                    // the previous tile's output phase has read its last stage out of this buffer
#include "pmr_fe_tile.inc"
    }
}

// ---------------------------------------------------------------------------------------------
// Level 2 of the two-level front end: 2048 samples of the decimated ring per tile -> m = 5 stage -> m = 10 stage -> resampler.
// The ring samples produced by THIS call still miss level 1's dc carry, V_c1 * K1 * mu^i' (k_fe_carry computed the V_c1 and
// already fixed the last few in place: index >= fix_limit); it is subtracted here while loading.  A tile starts at a multiple
// of 4 in absolute ring index, so samples are loaded as 16-byte pairs.  1/16 of the raw rate flows through here (cfg5).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 6) void k_fe_level2(pmr_fe_params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256;                                // 2048 ring samples per tile
    cf *buf = reinterpret_cast<cf *>(smem) + FE_PAD;       // input (layout L(8)), then z1 (L(4)), then z2 (L(2)), all in place
    cf *R0 = buf;
    const int tid = threadIdx.x;
    const int c = (int)pmr_xcd_contiguous(blockIdx.x, gridDim.x);
    const long b0 = (long)c * p.T_own - p.Hh - p.pend;     // index of tile sample 0 among this call's new ring samples

    const unsigned long long qa = (unsigned long long)c * p.TQ;
    const fe_arb_plan ap = fe_arb_prepare<NT>(p, qa, tid);
    float bk0[14], bk1[14];

    if (tid < FE_PAD) buf[tid - FE_PAD] = cfm(0.f, 0.f);
    {
        const cf *__restrict__ ring = (const cf *)p.in_ring;
        const cf *__restrict__ V1 = (const cf *)p.fixV;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = 2 * (tid + NT * k);                              // tile-local index of the pair's first sample
            const long long a = (long long)p.in_abs0 + b0 + i, jn = b0 + i;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a >= 0 && jn < (long long)p.n_in) v[k] = *reinterpret_cast<const float4 *>(ring + ((unsigned long long)a & p.in_mask));
        }
#pragma unroll
        for (int k = 0; k < 14; k++) { bk0[k] = ap.b0p[k]; bk1[k] = ap.b1p[k]; }
        // Level-1 carry of a new ring sample j (0 <= j < fix_limit): tile c1 = j / TQ by a float reciprocal with an exact fix-up
        // (j < 2^24), gain from ONE table.  The pair's second sample is the next tile-local index, or index 0 of the next tile.
        const unsigned TQ1 = p.fix_TQ;
        const auto locate = [&](unsigned j, unsigned &c1, unsigned &r) {
            c1 = (unsigned)((float)j * p.fix_rTQ);
            int rr = (int)(j - c1 * TQ1);
            if (rr < 0) { c1--; rr += (int)TQ1; } else if (rr >= (int)TQ1) { c1++; rr -= (int)TQ1; }
            r = (unsigned)rr;
        };
        const auto apply = [&](unsigned c1, unsigned r, float &re, float &im) {
            const float g = p.fix_G[r + p.fix_HhQ];
            const cf Vc = V1[c1];
            re = fmaf(-Vc.x, g, re); im = fmaf(-Vc.y, g, im);
        };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = 2 * (tid + NT * k);
            const long long jn = b0 + i;
            float4 w = v[k];
            if (jn + 1 >= (long long)p.n_in) { w.z = 0.f; w.w = 0.f; }    // the pair's second sample lies beyond the block
            if (V1 && jn + 1 >= 0 && jn < (long long)p.fix_limit) {       // the pair touches the range level 2 corrects
                unsigned c1, r;
                locate((unsigned)(jn < 0 ? 0 : jn), c1, r);
                if (jn >= 0) {
                    apply(c1, r, w.x, w.y);
                    if (++r == TQ1) { r = 0; c1++; }
                }
                if (jn + 1 < (long long)p.fix_limit) apply(c1, r, w.z, w.w);
            }
            cf *d = R0 + lidx<8>(i);                                       // the pair never straddles an 8-sample chunk
            d[0] = cfm(w.x, w.y);
            d[1] = cfm(w.z, w.w);
        }
    }
    __syncthreads();
    hb_stage_ip<4, 5>(R0, tid, NT, p.taps_k, 1.0f);                        // 1024 outputs, L(8) -> L(4), in place
    hb_stage_ip<2, 10>(R0, tid, NT, p.taps_k + 10, p.zeta);                //  512 outputs, L(4) -> L(2), in place
    fe_arb_store<NT, 1>(p, ap, qa, R0, bk0, bk1, tid);
}

// ---------------------------------------------------------------------------------------------
#ifndef FE_EXTRA_LDS
#define FE_EXTRA_LDS 0      /* experiment: bytes of unused LDS per workgroup on top of pmr_fe_params.lds_pad */
#endif
template <int MODE, int N3, int TAIL>
static int launch_fast(hipStream_t st, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev)
{
    const size_t lds = (FE_PAD + (N3 == 0 ? 4352 : 4096) + 4 + 40) * sizeof(cf) + FE_EXTRA_LDS + (MODE == FE_FULL ? p->lds_pad : 0u);
    auto kern = k_fe_fast<MODE, N3, TAIL>;
#ifdef FE_ANYORDER      /* experiment: AQL packet without the barrier bit -- this launch's workgroups may start under the previous one's tail */
    hipExtLaunchKernelGGL(kern, dim3(ntiles), dim3(256), (unsigned)lds, st, ev ? (hipEvent_t)ev->start : nullptr,
                          ev ? (hipEvent_t)ev->stop : nullptr, hipExtAnyOrderLaunch, *p);
#else
    PMR_LAUNCH_EV(kern, dim3(ntiles), dim3(256), lds, st, ev, *p);
#endif
    return (int)hipGetLastError();
}


template <int MODE, int N3, int TAIL>
static int launch_loop(hipStream_t st, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev)
{
    const size_t lds = (FE_PAD + (N3 == 0 ? 4352 : 4096) + 4 + 40) * sizeof(cf) + FE_EXTRA_LDS + (MODE == FE_FULL ? p->lds_pad : 0u);
    /* the grid is a whole number of "rounds" of the 4 x 256 workgroups the chip holds at a time: with any other count the last round
     * runs on a mostly empty chip (tpw = 16 as ceil(17007 / 16) = 1063 workgroups took 127 us: 1024 workgroups walk 16 tiles, then 39
     * walk 16 more) */
    const unsigned slots = 4u * 256u;
    unsigned rounds = (ntiles + (p->tiles_per_wg * slots) / 2) / (p->tiles_per_wg * slots);
    if (rounds < 1) rounds = 1;
    const unsigned nwg = rounds * slots < ntiles ? rounds * slots : ntiles;
    auto kern = k_fe_loop<MODE, N3, TAIL>;
    PMR_LAUNCH_EV(kern, dim3(nwg), dim3(256), lds, st, ev, *p, ntiles);
    return (int)hipGetLastError();
}

/* the specialised kernels cover: N3 six-tap stages, then optionally (m = 5, m = 10); 256 x 16 tiles */
static int fast_pattern_m(const int *m, int h, int *n3, int *tail)
{
    int k = 0;
    while (k < h && m[k] == 3) k++;
    *n3 = k;
    if (k == h) { *tail = 0; return k >= 1; }
    if (k + 2 == h && m[k] == 5 && m[k + 1] == 10) { *tail = 1; return 1; }
    return 0;
}
static int fast_pattern(const pmr_fe_params *p, int *n3, int *tail) { return fast_pattern_m(p->m, p->h, n3, tail); }

extern "C" int pmr_fe_fast_covers(int mode, const int *m, int h)
{
    int n3 = 0, tail = 0;
    if (!fast_pattern_m(m, h, &n3, &tail)) return 0;
    if (mode == FE_FULL) return tail && n3 <= 3;
    if (mode == FE_L1) return !tail && n3 >= 2 && n3 <= 5;
    return 0;
}

extern "C" int pmr_launch_fe_fast(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev)
{
    hipStream_t st = (hipStream_t)s;
    int n3 = 0, tail = 0;
    if (!p->taps_valid || !fast_pattern(p, &n3, &tail)) return -1;
    if (p->tiles_per_wg > 1 && p->in_fmt == 0 && n3 >= 1) {    /* blocks of thousands of tiles: a workgroup walks tiles_per_wg of them */
        if (p->mode == FE_FULL && tail) {
            if (n3 == 1) return launch_loop<FE_FULL, 1, 1>(st, p, ntiles, ev);
            if (n3 == 2) return launch_loop<FE_FULL, 2, 1>(st, p, ntiles, ev);
            if (n3 == 3) return launch_loop<FE_FULL, 3, 1>(st, p, ntiles, ev);
        }
        if (p->mode == FE_L1 && !tail) {
            if (n3 == 2) return launch_loop<FE_L1, 2, 0>(st, p, ntiles, ev);
            if (n3 == 3) return launch_loop<FE_L1, 3, 0>(st, p, ntiles, ev);
            if (n3 == 4) return launch_loop<FE_L1, 4, 0>(st, p, ntiles, ev);
            if (n3 == 5) return launch_loop<FE_L1, 5, 0>(st, p, ntiles, ev);
        }
    }
    if (p->mode == FE_FULL && tail) {
        if (n3 == 0) return launch_fast<FE_FULL, 0, 1>(st, p, ntiles, ev);
        if (n3 == 1) return launch_fast<FE_FULL, 1, 1>(st, p, ntiles, ev);
        if (n3 == 2) return launch_fast<FE_FULL, 2, 1>(st, p, ntiles, ev);
        if (n3 == 3) return launch_fast<FE_FULL, 3, 1>(st, p, ntiles, ev);
    }
    if (p->mode == FE_L1 && !tail) {
        if (n3 == 2) return launch_fast<FE_L1, 2, 0>(st, p, ntiles, ev);
        if (n3 == 3) return launch_fast<FE_L1, 3, 0>(st, p, ntiles, ev);
        if (n3 == 4) return launch_fast<FE_L1, 4, 0>(st, p, ntiles, ev);
        if (n3 == 5) return launch_fast<FE_L1, 5, 0>(st, p, ntiles, ev);
    }
    return -1;
}

extern "C" int pmr_launch_fe_level2_fast(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles)
{
    if (!ntiles) return 0;
    const size_t lds = (FE_PAD + (2048 + 256)) * sizeof(cf);          /* 18.9 KB: the stages run in place */
    PMR_KLAUNCH(k_fe_level2, dim3(ntiles), dim3(256), lds, (hipStream_t)s, *p);
    return (int)hipGetLastError();
}
