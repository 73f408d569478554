// pmr_fe_fast.hip -- the SPECIALISED front-end kernels for gfx950: dc-block -> half-band cascade -> arbitrary resampler
// (reference src/sdr_pmr446.c:795-796: iirfilt_crcf_execute_block + msresamp_crcf_execute) for the cascades liquid's
// msresamp2 design produces at stop-bands of 50 ... 72 dB: N3 six-tap stages (m = 3), then two longer stages (MA, MB) = (5, 10) at the
// reference's As = 60 dB (:426) -- (4, 8), (5, 9), (5, 11), (6, 11), (6, 12) for the designs around it (SURVEY A.3 is confidence [M]
// on what liquid really picks: every pair gets the same kernels) --, then the 256 x 14 polyphase bank.
//
//   k_fe_fast<FE_FULL, N3, MA, MB>  whole front end in one pass over the raw block (cfg2: N3 = 1, cfg3: N3 = 2; N3 = 0: the reference's
//                                   own 1.024 MS/s plan, whose cascade is just the two long stages)
//   k_fe_fast<FE_L1,  N3, 0, 0>     level 1 of a deep cascade (cfg5 / dsd_in: N3 = 4): dc-block + N3 six-tap stages -> decimated ring
//   k_fe_level2<MA, MB>             level 2: ring in (level 1's dc carry applied at load) -> stage MA -> stage MB -> resampler
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
//   * (round 6) the stage behind the first one -- all remaining stages in level 1 of the deep cascades -- is computed straight from
//     registers: the left neighbours' samples come by DPP, only a wave's first lanes fetch the previous wave's last ones through a
//     few dozen samples of LDS (hb_stage_reg, pmr_fe_common.hpp); the later stages of the one-level kernels ping-pong between two
//     LDS regions inside the dead raw tile (one barrier per stage);
//   * (round 6) that level 1 needs exactly the raw tile's 32 768 bytes of LDS (its scratch lives in each wave's own quarter of the
//     tile): five tiles per CU; the one-level kernels 33.7 KB (zero pad in front of the tile + scratch behind it): four.
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


typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// slot of 16-byte chunk g (and chunk of slot g: the map is an involution)
static __device__ __forceinline__ int fe_swz(int g) { return (g & ~7) | ((g & 7) ^ ((g >> 4) & 7)); }

#ifdef FE_STAMP
// diagnostic build (tools/fe_phase_times.py): s_memtime at the phase boundaries of every tile, thread 0 -> fe_stamps[tile][8]
__device__ unsigned long long fe_stamps[65536 * 8];
#define FE_STAMP_AT(i) do { if (threadIdx.x == 0 && blockIdx.x < 65536) { fe_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();      \
        if ((i) == 0) fe_stamps[blockIdx.x * 8 + 5] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 32) |                 \
                                                      __builtin_amdgcn_s_getreg(4 | (31 << 11)); /* XCC_ID, HW_ID: which CU ran the tile */ } } while (0)
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

template <int MA, int MB>
static __device__ __forceinline__ void fe_level2_tile(const pmr_fe_params &p, const int c, char *smem, const int tid);

#ifdef EXP_L2_INLINE
/* TIMING EXPERIMENT (profiles/r06_ab_log.txt r6a; WRONG results): what could "level 2 inside the level-1 launch" buy?  The level-1 tile
 * that sits EXP_L2_INLINE tiles behind the last contributor of a level-2 tile runs that tile's body after its own ring stores -- no
 * counters, no acquire, the dc carries of the contributors read from whatever the carry buffer holds: everything a real
 * implementation must ADD is left out, so the build bounds its gain from above.  -DEXP_L2_INLINE_ATOMIC adds the cheapest
 * conceivable hand-shake: every tile drains its stores and makes ONE agent-scope atomic add (no return value) on the worker's
 * counter, the worker reads that counter once. */
#define FE_P2_PARAM , pmr_fe_params p2, unsigned long long *l2cnt
#else
#define FE_P2_PARAM
#endif

// Level 1 of the deep cascades with every stage in registers (ALL_REG) needs no LDS beyond the raw tile: no stage window reaches left
// of the tile (no zero pad), and the scan / halo scratch of wave w lives in wave w's OWN quarter of the tile, which only that wave
// reads and which is dead once its threads hold their samples.  32 768 bytes exactly: FIVE tiles per CU (5 x 32 KB = the whole 160 KB)
// instead of four at 33.7 KB -- a quarter more bytes in flight per CU.
template <int MODE, int N3, int MA>
static constexpr bool fe_tight_lds()
{
#if defined(FE_LAST_LDS) || defined(FE_L1_LDS23) || defined(FE_S1_LDS) || defined(FE_NO_TIGHT)
    return false;
#else
    return MODE == FE_L1 && N3 == 4 && MA == 0;
#endif
}

template <int MODE, int N3, int MA, int MB>
__global__ __launch_bounds__(256, 4) void k_fe_fast(pmr_fe_params p FE_P2_PARAM)
{
    constexpr int TAIL = MA > 0 ? 1 : 0;
    static_assert((MA > 0) == (MB > 0), "the two long stages come as a pair");
    static_assert(N3 >= 1 || (MODE == FE_FULL && TAIL == 1), "a cascade without six-tap stages is the two long stages");
    static_assert((4 * MB - 2) * 3 / 2 <= FE_PAD && (4 * MA - 2) * 3 / 2 <= FE_PAD, "zero pad in front of the tile: a stage's window left of sample 0");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 256, SPT = 16, N0 = NT * SPT;
    constexpr int H = N3 + 2 * TAIL;
    // stage 1 (execution index 1: a six-tap stage when N3 >= 2, else the first long stage) computed from registers, see phase B
    constexpr int M1 = N3 >= 2 ? 3 : (MA > 0 ? MA : 3);
#ifdef FE_S1_LDS            /* A/B hook: stage 1 through LDS like the later stages (rounds 2-5) */
    constexpr bool S1_REG = false;
#else
    constexpr bool S1_REG = N3 >= 1 && H >= 2;
#endif
#if defined(FE_LAST_LDS) || defined(FE_L1_LDS23)   /* A/B hooks: level 1's stages 2 (and 3: FE_LAST_LDS) through LDS, as in rounds 4-5 */
    constexpr bool ALL_REG = false;
#else
    constexpr bool ALL_REG = S1_REG && MODE == FE_L1 && N3 == 4 && !TAIL;
#endif
    cf ylast = cfm(0.f, 0.f);                              // level 1's last-stage output of this thread (LAST_IN_REG / ALL_REG)
    constexpr int R1_OFF = (N0 / 2) + (N0 / 2) / 8;         // z1 (2048 samples, layout L(8)) fills [0, R1_OFF) of the tile area
    // LDS: [FE_PAD zero pad | raw tile, 4096 samples = 2048 swizzled 16-byte chunks; afterwards R0 (z3 ..., 2304 slots: z1 itself stays in
    // registers since round 6) and R1 (z2, 1280 slots) | scan scratch] = 33.7 KB -> four tiles per CU; TIGHT (below): the raw tile only
    // N3 == 0 (no six-tap stage: the reference's own 1.024 MS/s plan is m = 5, 10): the dc-blocked tile goes back to LDS in
    // layout L(16) (4352 slots) and the m = 5 stage runs from there in place; the scan scratch sits behind it
    constexpr int SCR = N3 == 0 ? N0 + N0 / 16 : N0;
    constexpr bool TIGHT = fe_tight_lds<MODE, N3, MA>();
    cf *buf = reinterpret_cast<cf *>(smem) + (TIGHT ? 0 : FE_PAD);
    // scan aggregate of wave w: wagg[w * WST]; halo of stage 0 across the wave boundary: bnd[w * BST + 0 .. 9].  TIGHT: inside wave w's own
    // quarter of the tile (1024 samples per wave), otherwise behind the tile
    constexpr int WST = TIGHT ? 1024 : 1, BST = TIGHT ? 1024 : 10;
    cf *wagg = TIGHT ? buf : buf + SCR;
    cf *bnd = TIGHT ? buf + 8 : wagg + NT / 64;               // [4][10]
    cf *xch = TIGHT ? buf + 32 : buf;                         // exchange areas of the register stages (hb_stage_reg): 96 + 48 + 40 samples

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
    const long b0 = (long)c * p.T_own - p.Hh - p.pend;     // block-relative index of tile sample 0

    // ---- phase A: raw tile -> LDS ----
    if constexpr (!TIGHT) { if (tid < FE_PAD) buf[tid - FE_PAD] = cfm(0.f, 0.f); }
    const bool fast = p.in_fmt == 0 && b0 >= 0 && b0 + N0 <= (long)p.n_in && ((reinterpret_cast<uintptr_t>(x + b0) & 15) == 0);
    if (fast) {
        // wave-instruction i of wave w moves slots s0 .. s0 + 63, s0 = 512 w + 64 i (1 KiB).  fe_swz(s0 + lane) - s0 depends on
        // the parity of i only ((s0 >> 4) & 7 = 4 (i & 1)), so the source address is a UNIFORM base (scalar registers, immediate
        // i * 1 KiB) plus one of two per-lane byte offsets, and the LDS base (M0) is scalar too: no per-instruction vector
        // address arithmetic (it was ~5 VALU per DMA instruction)
        const int ws = __builtin_amdgcn_readfirstlane(wave);
        const char *srcw = reinterpret_cast<const char *>(x + b0) + (size_t)ws * (N0 / 8) * 16;
        float4 *ldsw = reinterpret_cast<float4 *>(buf) + ws * (N0 / 8);
        const unsigned sw0 = (unsigned)((lane & ~7) | ((lane & 7) ^ ((lane >> 4) & 7))) * 16u;
        const unsigned sw1 = (unsigned)((lane & ~7) | ((lane & 7) ^ (((lane >> 4) + 4) & 7))) * 16u;
#pragma unroll
        for (int i = 0; i < N0 / 2 / NT; i++)
            __builtin_amdgcn_global_load_lds((gptr_t *)(srcw + i * 1024 + ((i & 1) ? sw1 : sw0)), (lptr_t *)(ldsw + i * 64), 16, 0,
                                             FE_DMA_AUX);
    } else if (p.in_fmt == 0) {
        // edge tiles (history before the block, zeros beyond it, or an unaligned block): plain loads into the same image
#pragma unroll 4
        for (int i = tid; i < N0; i += NT) {
            const long b = b0 + i;
            cf w = cfm(0.f, 0.f);
            if (b < 0) { const long hi = (long)p.hcap + b; if (hi >= 0) w = hist[hi]; }
            else if (b < (long)p.n_in) w = x[b];
            buf[2 * fe_swz(i >> 1) + (i & 1)] = w;
        }
    } else {
        // integer sample formats, converted on the way in (synchronous zero-copy calls: the block sits in pinned HOST memory, 2 or
        // 4 bytes per sample cross the link instead of 8).  Four samples per thread and step: one 8- / 16-byte request where the
        // group lies inside the block and is naturally aligned, else sample by sample
        const int bps = p.in_fmt == 1 ? 4 : 2;
#pragma unroll 2
        for (int i4 = tid; i4 < N0 / 4; i4 += NT) {
            const int i = 4 * i4;
            const long b = b0 + i;
            cf w[4];
            const char *src = reinterpret_cast<const char *>(p.x) + b * bps;
            if (b >= 0 && b + 4 <= (long)p.n_in && ((reinterpret_cast<uintptr_t>(src) & (uintptr_t)(4 * bps - 1)) == 0)) {
                if (p.in_fmt == 1) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(src);
                    const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        w[k] = cfm((float)(short)(vv[k] & 0xffffu) * (1.0f / 32768.0f), (float)(short)(vv[k] >> 16) * (1.0f / 32768.0f));
                } else {
                    const uint2 v = *reinterpret_cast<const uint2 *>(src);
                    const unsigned vv[2] = {v.x, v.y};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned h2 = (vv[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                        w[k] = cfm(((float)(h2 & 0xffu) - 127.5f) * (1.0f / 127.5f), ((float)(h2 >> 8) - 127.5f) * (1.0f / 127.5f));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const long bk = b + k;
                    w[k] = cfm(0.f, 0.f);
                    if (bk < 0) { const long hi = (long)p.hcap + bk; if (hi >= 0) w[k] = hist[hi]; }
                    else if (bk < (long)p.n_in) w[k] = fe_raw(p.x, bk, p.in_fmt);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) buf[2 * fe_swz((i + k) >> 1) + ((i + k) & 1)] = w[k];
        }
    }

    // ---- everything that comes from tables, requested while the tile streams in ----
    const unsigned long long qa = (unsigned long long)c * p.TQ;
    fe_arb_plan ap;
    float bk0[14], bk1[14];
    if constexpr (MODE == FE_FULL) {
        ap = fe_arb_prepare<NT, true>(p, qa, tid);
        if (p.tile_j && tid == 0) { ((unsigned long long *)p.tile_j)[2 * c] = ap.ja; ((unsigned long long *)p.tile_j)[2 * c + 1] = ap.jb; }
#pragma unroll
        for (int k = 0; k < 14; k++) { bk0[k] = ap.b0p[k]; bk1[k] = ap.b1p[k]; }
    }
    const float lp = p.lam_lane_pow[lane], l15 = p.lam_lane_pow[(lane & 15) + 1], l31 = p.lam_lane_pow[(lane & 31) + 1];
    __syncthreads();                                       // (the compiler drains the DMA before the barrier)
    FE_STAMP_AT(1);
    // (a wave-local wait instead -- each wave's DMA fetches exactly the chunks its own threads read back -- was measured: neutral
    //  in all three plans, round 3; and a build of this kernel with 101 instead of 88 VGPRs cost the cfg2 chain 10 %: the audio
    //  FIR's and the channelizer's waves no longer fit beside four of these tiles on a SIMD, so keep an eye on the register count)

    cf xs[SPT];
    {
        const float4 *rb = reinterpret_cast<const float4 *>(buf) + 8 * tid;
        const int sw = (tid >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 v = rb[j ^ sw];
            xs[2 * j] = cfm(v.x, v.y); xs[2 * j + 1] = cfm(v.z, v.w);
        }
    }

    // ---- phase B: dc blocker (:795) from ZERO state + first (six-tap) stage straight from registers ----
    {
        cf yb[SPT];
        cf v = cfm(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SPT; j++) v = cfma(lam, v, xs[j]);             // v0 = x - a1 v1
        // inclusive decayed scan across the wave (DPP): inc_l = sum_{s<=l} lambda^(SPT (l-s)) agg_s
        v = cfma(p.lam_pow16[0], dpp0c<0x111>(v), v);                      // row_shr:1
        v = cfma(p.lam_pow16[1], dpp0c<0x112>(v), v);                      // row_shr:2
        v = cfma(p.lam_pow16[2], dpp0c<0x114>(v), v);                      // row_shr:4
        v = cfma(p.lam_pow16[3], dpp0c<0x118>(v), v);                      // row_shr:8
        v = cfma(l15, dpp0c<0x142, 0xA>(v), v);                            // row_bcast:15 -> rows 1, 3
        v = cfma(l31, dpp0c<0x143, 0xC>(v), v);                            // row_bcast:31 -> rows 2, 3
        if (lane == 63) wagg[wave * WST] = v;
        const cf ex = dpp0c<0x138>(v);                                     // wave_shr:1 (lane 0 <- 0)
        __syncthreads();                                                   // also: every thread holds its raw samples
        // v (local) at the end of the previous wave: Horner over the aggregates of the waves before this one.  Branch-free (the
        // three reads issued together, each step selected by the wave index): as a loop over `wave` it compiled to an exec-masked
        // loop with one dependent LDS read per trip -- ~400 cycles for wave 3, which the whole tile waits for at the next barrier
        cf cw = cfm(0.f, 0.f);
        {
            const cf a0 = wagg[0], a1 = wagg[WST], a2 = wagg[2 * WST];
            const cf c1 = cfma(p.lam_wave, cw, a0);
            cw = wave > 0 ? c1 : cw;
            const cf c2 = cfma(p.lam_wave, cw, a1);
            cw = wave > 1 ? c2 : cw;
            const cf c3 = cfma(p.lam_wave, cw, a2);
            cw = wave > 2 ? c3 : cw;
        }
        cf v1 = cfma(lp, cw, ex);
        // stray probes (block start - 1 in tile 0, block end in the last tile) sit at arbitrary offsets
        const int pL = (c == 0) ? p.Hh + p.pend - 1 : -1;
        const int pE = (c == p.c_end) ? p.off_end : -1;
        const cf v1s = v1;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const cf v0 = cfma(lam, v1, xs[j]);
            yb[j] = csub(v0, v1);                                          // y = v0 - v1
            v1 = v0;
        }
        if (pL >= 0 || pE >= 0) {
            // only the block's first and last tile get here (uniform branch): the one thread that owns a stray probe runs its
            // recurrence again.  Inside the loop above the test was an exec-mask branch per sample -- sixteen taken branches in
            // the middle of the kernel's longest dependent chain, in every tile
            if ((pL >= 0 && pL / SPT == tid) || (pE >= 0 && pE / SPT == tid)) {
                cf u = v1s;
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    u = cfma(lam, u, xs[j]);
                    if (SPT * tid + j == pL) ((cf *)p.probeL)[0] = u;
                    if (SPT * tid + j == pE) ((cf *)p.probeE)[0] = u;
                }
            }
        }
        if (tid == p.Hh / SPT - 1) ((cf *)p.probeA)[c] = v1;      // local v at tile offset Hh-1
        if (tid == NT - 1) ((cf *)p.probeB)[c] = v1;              // local v at tile offset N0-1
        if constexpr (N3 == 0) {
            // no six-tap stage: dc-blocked samples -> LDS, layout L(16) (the raw tile is dead: every thread holds its samples)
            cf *o = buf + tid * 17;
#pragma unroll
            for (int i = 0; i < SPT; i++) o[i] = yb[i];
        } else {
        // halo of stage 0: the previous thread's yb[6..15] (lane 0: previous wave's lane 63, through LDS)
        cf W[26];
#pragma unroll
        for (int i = 0; i < 10; i++) W[i] = dpp0c<0x138>(yb[6 + i]);
#pragma unroll
        for (int i = 0; i < 16; i++) W[10 + i] = yb[i];
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < 10; i++) bnd[wave * BST + i] = yb[6 + i];
        }
        __syncthreads();
        if (lane == 0 && wave > 0) {
#pragma unroll
            for (int i = 0; i < 10; i++) W[i] = bnd[(wave - 1) * BST + i];
        }
        // z1[8 tid + q] = W[2q + 5] + sum_j h1[j] W[2q + 2j]   (window offset 0 <-> sample 16 tid - 10)
        const float scale0 = H == 1 ? p.zeta : 1.0f;
        cf z[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            cf a = cfm(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 6; j++) a = cfma(p.taps_k[j], W[2 * q + 2 * j], a);
            z[q] = cadd_scale(W[2 * q + 5], a, scale0);
        }
        if constexpr (!S1_REG) {
            cf *o = buf + tid * 9;                         // z1 in layout L(8), region R0 (over the raw tile)
#pragma unroll
            for (int q = 0; q < 8; q++) o[q] = z[q];
        } else {
            // STAGE 1 STRAIGHT FROM REGISTERS (round 6, hb_stage_reg): thread t holds z1[8t .. 8t + 7]; the elements of the threads to its
            // left that its four outputs need come by DPP, a wave's first lanes get theirs through 96 samples of LDS (the raw tile is
            // dead: every thread holds its samples since the scan's barrier) -- instead of a write of all of z1 to LDS and 13 - 17
            // window reads back per thread.  Same operations in the same order as hb_stage_pp<4, M1>: bit-identical (tools/pcm_hash.py).
            cf y1[4];
            hb_stage_reg<8, M1>(z, y1, xch, lane, wave, p.taps_k + 6, H == 2 ? p.zeta : 1.0f);
            if constexpr (ALL_REG) {
                // level 1 of the deep cascades (cfg5, dsd_in): the remaining two six-tap stages the same way -- no stage of this kernel
                // goes through LDS any more, only 184 boundary samples per tile do
                cf y2[2], y3[1];
                hb_stage_reg<4, 3>(y1, y2, xch + 96, lane, wave, p.taps_k + 12, 1.0f);
                hb_stage_reg<2, 3>(y2, y3, xch + 96 + 48, lane, wave, p.taps_k + 18, p.zeta);
                ylast = y3[0];
            } else {
                // (no barrier in front: the exchange area lies in R0's first slots, the outputs go to R1; R0 is not written before the barrier below)
                cf *o = buf + R1_OFF + tid * 5;                                 // stage 1's outputs, layout L(4), region R1 (as hb_stage_pp<4, .> leaves them)
#pragma unroll
                for (int pp = 0; pp < 4; pp++) o[pp] = y1[pp];
            }
        }
    }
        }
    if constexpr (!ALL_REG)
    __syncthreads();
    FE_STAMP_AT(2);

    // ---- phase C: remaining stages, ping-pong R0 <-> R1; stage e (execution index) has 2048 >> e outputs ----
    cf *R0 = buf, *R1 = buf + R1_OFF;
#define FE_STAGE(E, MM, TOFF) do { constexpr int NOUT = (N0 / 2) >> (E); constexpr int PP = NOUT >= NT ? NOUT / NT : 1;          \
        hb_stage_pp<PP, MM>(((E) & 1) ? R0 : R1, ((E) & 1) ? R1 : R0, tid, NOUT / PP, p.taps_k + (TOFF),                         \
                            (E) == H - 1 ? p.zeta : 1.0f); } while (0)
    // six-tap stages 1 .. N3-1 (taps at 6 e), then the two long stages (m = MA, MB: 5 and 10 in the reference's design)
    if constexpr (N3 >= 2 && !S1_REG) FE_STAGE(1, 3, 6);
    if constexpr (N3 >= 3 && !ALL_REG) FE_STAGE(2, 3, 12);
    // level 1 with four six-tap stages (cfg5, dsd_in): the last stage has one output per thread -- it stays in a register and goes
    // straight to the ring (below): one LDS write, one barrier and two LDS reads less at the end of the tile's life
#ifdef FE_LAST_LDS
    constexpr bool LAST_IN_REG = false;
#else
    constexpr bool LAST_IN_REG = MODE == FE_L1 && N3 == 4 && !TAIL;
#endif
    if constexpr (ALL_REG) { }
    else if constexpr (LAST_IN_REG) ylast = hb_stage_out1<3>(R0, tid, p.taps_k + 18, p.zeta);
    else if constexpr (N3 >= 4) FE_STAGE(3, 3, 18);
    if constexpr (N3 >= 5) FE_STAGE(4, 3, 24);
    if constexpr (TAIL && N3 == 0) hb_stage_ip<8, MA>(buf, tid, NT, p.taps_k, 1.0f);           // 2048 outputs, L(16) -> L(8), in place
    else if constexpr (TAIL && !(S1_REG && N3 == 1)) FE_STAGE(N3, MA, 6 * N3);
    if constexpr (TAIL) FE_STAGE(N3 + 1, MB, 6 * N3 + 2 * MA);
#undef FE_STAGE
    constexpr int NLAST = (N0 / 2) >> (H - 1);
    constexpr int PLAST = H == 1 ? 8 : (NLAST >= NT ? NLAST / NT : 1);
    constexpr int GS = PLAST >= 8 ? 3 : (PLAST == 4 ? 2 : 1);                   // final layout L(1 << GS)
    const cf *fin = ((H - 1) & 1) ? R1 : R0;                                      // stage e writes R1 when e is odd
    FE_STAMP_AT(3);

    if constexpr (LAST_IN_REG) {
        // thread t holds output t of the last stage; outputs HhQ .. HhQ + nown - 1 are the tile's own.  A lane at an even ring position
        // stores its sample and its right neighbour's (DPP wave_shl:1) as 16 bytes; lane 63 of a wave, the tile's last sample and the
        // odd head sample go alone, and so does lane 0 when its left partner sits in the previous wave
        cf *__restrict__ out = (cf *)p.out;
        const int nown = (int)((qa + p.TQ <= p.Q) ? p.TQ : (p.Q > qa ? p.Q - qa : 0));   // samples this tile stores
        const cf ynext = dpp0c<0x130>(ylast);                                            // wave_shl:1 (lane 63 <- 0)
        const int i = tid - p.HhQ;
        if (i >= 0 && i < nown) {
            const unsigned long long r = p.out_pos0 + qa + (unsigned)i;
            cf *o = out + (r & (p.out_mask & FE_OUT_AND));
            if (!(r & 1ull)) {
                if (i + 1 < nown && lane != 63) FE_STORE4(o, make_float4(ylast.x, ylast.y, ynext.x, ynext.y));
                else *o = ylast;
            } else if (i == 0 || lane == 0) {
                *o = ylast;
            }
        }
    } else if constexpr (MODE == FE_L1) {
        // pairs of adjacent samples per lane through 16-byte stores (8-byte stores run at ~0.6x the rate); pairs start at
        // even ring positions, so they are aligned and never straddle the ring end
        cf *__restrict__ out = (cf *)p.out;
        const int nown = (int)((qa + p.TQ <= p.Q) ? p.TQ : (p.Q > qa ? p.Q - qa : 0));   // samples this tile stores
        const auto ld = [&](int i) { return fin[(p.HhQ + i) + ((p.HhQ + i) >> GS)]; };
        const int head = (int)((p.out_pos0 + qa) & 1ull) && nown > 0;
        if (head && tid == 0) out[(p.out_pos0 + qa) & (p.out_mask & FE_OUT_AND)] = ld(0);
        const int npair = (nown - head) >> 1;
        for (int t = tid; t < npair; t += NT) {
            const int i = head + 2 * t;
            const cf a = ld(i), b = ld(i + 1);
            FE_STORE4(out + ((p.out_pos0 + qa + i) & (p.out_mask & FE_OUT_AND)), make_float4(a.x, a.y, b.x, b.y));
        }
        if (((nown - head) & 1) && tid == 0) out[(p.out_pos0 + qa + nown - 1) & (p.out_mask & FE_OUT_AND)] = ld(nown - 1);
    } else {
        fe_arb_store<NT, GS>(p, ap, qa, fin, bk0, bk1, tid);
    }
    FE_STAMP_AT(4);
    // ---- raw history for the next call (last hcap samples of old history || block), by tile 0 ----
    if (c == 0 && p.new_hist) {
        cf *__restrict__ nh = (cf *)p.new_hist;
        for (int i = tid; i < p.hcap; i += NT) {
            const long sb = (long)i + (long)p.n_in - (long)p.hcap;      // block-relative index
            nh[i] = sb < 0 ? hist[(long)i + p.n_in] : fe_raw(p.x, sb, p.in_fmt);
        }
    }
#ifdef EXP_L2_INLINE
    if constexpr (MODE == FE_L1 && N3 == 4) {
        const int per = (int)(gridDim.x >> 3);
        const int lc = per ? c % per : c;                                 // position inside the XCD's contiguous range
#ifdef EXP_L2_INLINE_ATOMIC
        {
            // this tile's arrival: stores drained, then ONE agent-scope add on the counter of the level-2 tile its samples end in
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const long je = ((long)c + 1) * p.TQ - 1;
            long c2a = (je + p2.Hh + p2.pend) / p2.T_own;
            if (tid == 0) __hip_atomic_fetch_add(l2cnt + (c2a & 4095), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        if (lc >= EXP_L2_INLINE && c < (int)gridDim.x) {
            // level-2 tile c2 whose LAST input sample lies in the ring range tile c - LAG produced
            const long cl = (long)c - EXP_L2_INLINE;
            const long lo = cl * p.TQ, hi = lo + p.TQ;                     // new-sample indices tile cl stored
            long c2 = (lo - 2047 + p2.Hh + p2.pend + p2.T_own - 1) / p2.T_own;
            if (c2 < 0) c2 = 0;
            const long jl = c2 * p2.T_own - p2.Hh - p2.pend + 2047;
            if (jl >= lo && jl < hi && c2 <= p2.c_end) {
#ifdef EXP_L2_INLINE_ATOMIC
                if (tid == 0) {
                    const unsigned long long seen = __hip_atomic_load(l2cnt + (c2 & 4095), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen == 0xffffffffffffffffull) l2cnt[4095] = 1;    // (keeps the load alive; never true)
                }
#endif
                __syncthreads();                                           // every wave is done with the level-1 tile's LDS
                fe_level2_tile<5, 10>(p2, (int)c2, smem, tid);
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Level 2 of the two-level front end: 2048 samples of the decimated ring per tile -> stage MA -> stage MB -> resampler.
// The ring samples produced by THIS call still miss level 1's dc carry, V_c1 * K1 * mu^i' (k_fe_carry computed the V_c1 and
// already fixed the last few in place: index >= fix_limit); it is subtracted here while loading.  A tile starts at a multiple
// of 4 in absolute ring index, so samples are loaded as 16-byte pairs.  1/16 of the raw rate flows through here (cfg5).
// ---------------------------------------------------------------------------------------------
// one level-2 tile (c = its index): the body of k_fe_level2; 256 threads, (FE_PAD + 2304) cf of LDS from `smem`
template <int MA, int MB>
static __device__ __forceinline__ void fe_level2_tile(const pmr_fe_params &p, const int c, char *smem, const int tid)
{
    constexpr int NT = 256;                                // 2048 ring samples per tile
    cf *buf = reinterpret_cast<cf *>(smem) + FE_PAD;       // input (layout L(8)), then z1 (L(4)), then z2 (L(2)), all in place
    cf *R0 = buf;
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
#ifdef EXP_L2_NOFIX      /* timing experiment: level 2 without its carry arithmetic / table loads.  WRONG results */
            if (false) {
#else
            if (V1 && jn + 1 >= 0 && jn < (long long)p.fix_limit) {       // the pair touches the range level 2 corrects
#endif
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
#ifdef EXP_L2_STOP1      /* timing experiment: level 2 = loads + staging only */
    if (p.n_in != 0xffffffffu) return;
#endif
    hb_stage_ip<4, MA>(R0, tid, NT, p.taps_k, 1.0f);                       // 1024 outputs, L(8) -> L(4), in place
    hb_stage_ip<2, MB>(R0, tid, NT, p.taps_k + 2 * MA, p.zeta);            //  512 outputs, L(4) -> L(2), in place
    fe_arb_store<NT, 1>(p, ap, qa, R0, bk0, bk1, tid);
}

template <int MA, int MB>
__global__ __launch_bounds__(256, MB <= 10 ? 6 : 5) void k_fe_level2(pmr_fe_params p)      // (80 VGPRs; the longer pairs would spill: 96)
{
    static_assert((4 * MB - 2) * 5 / 4 <= FE_PAD, "zero pad in front of the tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fe_level2_tile<MA, MB>(p, (int)pmr_xcd_contiguous(blockIdx.x, gridDim.x), smem, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
#ifdef EXP_L2_INLINE
static pmr_fe_params g_exp_p2;                      /* level-2 parameters of the block whose level 1 is launched next (one thread: timing only) */
static unsigned long long *g_exp_cnt;
extern "C" void pmr_exp_set_l2_params(const pmr_fe_params *p2)
{
    g_exp_p2 = *p2;
    if (!g_exp_cnt) { (void)hipMalloc((void **)&g_exp_cnt, 4096 * sizeof(unsigned long long)); (void)hipMemset(g_exp_cnt, 0, 4096 * 8); }
}
#define FE_P2_ARG , g_exp_p2, g_exp_cnt
#else
#define FE_P2_ARG
#endif

template <int MODE, int N3, int MA, int MB>
static int launch_fast(hipStream_t st, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev)
{
#ifndef FE_EXTRA_LDS
#define FE_EXTRA_LDS 0      /* experiment: bytes of unused LDS per workgroup on top of pmr_fe_params.lds_pad */
#endif
    const size_t lds = fe_tight_lds<MODE, N3, MA>() ? 4096 * sizeof(cf) + FE_EXTRA_LDS
                     : (FE_PAD + (N3 == 0 ? 4352 : 4096) + 4 + 40) * sizeof(cf) + FE_EXTRA_LDS + (MODE == FE_FULL ? p->lds_pad : 0u);
    auto kern = k_fe_fast<MODE, N3, MA, MB>;
    PMR_LAUNCH_EV(kern, dim3(ntiles), dim3(256), lds, st, ev, *p FE_P2_ARG);
    return (int)hipGetLastError();
}

/* The (MA, MB) pairs the kernels are instantiated for: what liquid's msresamp2 design yields for the two stages next to the output
 * at stop-bands of 50 ... 72 dB (pmr_design.c; As = 60 -- the reference, :426 -- gives (5, 10)).  tools/ab_libs.py --resamp-as,
 * tests/test_gpu_parity.py (As parameter). */
#define FE_TAIL_PAIRS(X) X(4, 8) X(5, 9) X(5, 10) X(5, 11) X(6, 11) X(6, 12)
static int tail_pair_ok(int ma, int mb)
{
#define X(A, B) if (ma == A && mb == B) return 1;
    FE_TAIL_PAIRS(X)
#undef X
    return 0;
}

/* the specialised kernels cover: N3 six-tap stages, then optionally one of the pairs above; 256 x 16 tiles */
static int fast_pattern_m(const int *m, int h, int *n3, int *ma, int *mb)
{
    int k = 0;
    while (k < h && m[k] == 3) k++;
    *n3 = k; *ma = 0; *mb = 0;
    if (k == h) return k >= 1;
    if (k + 2 == h && tail_pair_ok(m[k], m[k + 1])) { *ma = m[k]; *mb = m[k + 1]; return 1; }
    return 0;
}

extern "C" int pmr_fe_fast_covers(int mode, const int *m, int h)
{
    int n3 = 0, ma = 0, mb = 0;
    if (!fast_pattern_m(m, h, &n3, &ma, &mb)) return 0;
    if (mode == FE_FULL) return ma && n3 <= 3;
    if (mode == FE_L1) return !ma && n3 >= 2 && n3 <= 5;
    if (mode == 2) return ma && n3 == 0;             /* level 2: the pair alone */
    return 0;
}

template <int MA, int MB>
static int launch_full(hipStream_t st, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev, int n3)
{
    if (n3 == 0) return launch_fast<FE_FULL, 0, MA, MB>(st, p, ntiles, ev);
    if (n3 == 1) return launch_fast<FE_FULL, 1, MA, MB>(st, p, ntiles, ev);
    if (n3 == 2) return launch_fast<FE_FULL, 2, MA, MB>(st, p, ntiles, ev);
    if (n3 == 3) return launch_fast<FE_FULL, 3, MA, MB>(st, p, ntiles, ev);
    return -1;
}

extern "C" int pmr_launch_fe_fast(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev)
{
    hipStream_t st = (hipStream_t)s;
    int n3 = 0, ma = 0, mb = 0;
    if (!p->taps_valid || !fast_pattern_m(p->m, p->h, &n3, &ma, &mb)) return -1;
    if (p->mode == FE_FULL && ma) {
#define X(A, B) if (ma == A && mb == B) return launch_full<A, B>(st, p, ntiles, ev, n3);
        FE_TAIL_PAIRS(X)
#undef X
    }
    if (p->mode == FE_L1 && !ma) {
        if (n3 == 2) return launch_fast<FE_L1, 2, 0, 0>(st, p, ntiles, ev);
        if (n3 == 3) return launch_fast<FE_L1, 3, 0, 0>(st, p, ntiles, ev);
        if (n3 == 4) return launch_fast<FE_L1, 4, 0, 0>(st, p, ntiles, ev);
        if (n3 == 5) return launch_fast<FE_L1, 5, 0, 0>(st, p, ntiles, ev);
    }
    return -1;
}

extern "C" int pmr_launch_fe_level2_fast(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles)
{
    if (!ntiles) return 0;
    if (p->h != 2 || !p->taps_valid) return -1;
    const size_t lds = (FE_PAD + (2048 + 256)) * sizeof(cf);          /* 19 KB: the stages run in place */
#define X(A, B) if (p->m[0] == A && p->m[1] == B) { PMR_KLAUNCH((k_fe_level2<A, B>), dim3(ntiles), dim3(256), lds, (hipStream_t)s, *p); return (int)hipGetLastError(); }
    FE_TAIL_PAIRS(X)
#undef X
    return -1;
}
