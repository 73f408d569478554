// pmr_ctcss.hip -- CTCSS tone detection for all M channels.
//
// reference: complementary low-pass branch  tmp1[k] = delay188(fm[k]) - hp[k]   src/sdr_pmr446.c:884-889
//            ctcss_execute(): dc-block (alpha 5e-4) :606, 38-tone Goertzel bank over CTCSS_BLOCK_SIZE = 2441
//            samples with the decision avg > 120 && max/avg > 10                  :366-409
//
// GPU formulation (everything per channel is linear, so time can be cut into pieces):
//  * the low-pass branch is ONE FIR with taps delta[d-188] - h[d]: the audio FIR kernel (pmr_fir_mfma4.hip /
//    k_fir_pair) run a second time on the discriminator ring, writing a time-major ring (done by the host);
//  * the dc-blocker v0 = x - a1 v1, y = v0 - v1 is a first-order linear scan, cut on the SAME segment grid as the Goertzel bank:
//    zero-state aggregates of every (segment, channel) (k_ct_seg_agg), a workgroup per channel strings them together with a
//    parallel scan of affine maps (k_ct_seg_scan), and the Goertzel kernel applies the exact recurrence to the segment it has
//    staged in LDS, from the segment's true carry -- no pass that rewrites the low-pass branch in memory, no latency-bound
//    serial loops (round 2's chunk scan walked 22 dependent loads per thread: 0.05 ms whatever the channel count);
//  * the Goertzel recurrence u0' = x + coef u0 - u1 has the impulse response U_n = sin((n+1)w)/sin(w), so after the N
//    samples of a block  u0 = sum_i x_i U_{N-1-i},  u1 = sum_i x_i U_{N-2-i}: a weighted sum that is split over 16 time
//    segments x 38 tones x M channels (k_ct_goertzel: LDS-tiled [16 channels] x [frames] . [frames] x [38 tones]) and reduced in a fixed order (k_ct_final), which also carries
//    the partial sums of a block that straddles two calls.  U is tabulated in double on the host.
//  * open-channel mask (the reference runs ctcss_execute for the squelch-selected channel only, :893): every kernel takes the
//    list of enabled channels (chan_list / n_chan; NULL = all M) and touches those channels only -- work, launch sizes and the
//    state that advances (dc-blocker state, Goertzel carry) are per OPEN channel; a closed channel's state stays as it was.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

// Segment grid shared by the dc-blocker scan and the Goertzel bank: block b = frames [b N, (b+1) N), cut into PMR_CT_SEG segments
// of SL = ceil(N / SEG) frames (the last one shorter); a call covers the segments that overlap [row0, row0 + ns), clipped.
struct ct_seg { long long lo; int len; unsigned n0; };             // first frame, frames inside the call, position inside the block
static __device__ __forceinline__ ct_seg ct_segment(long long b, unsigned seg, unsigned N, unsigned SL, long long row0, unsigned ns)
{
    long long lo = b * (long long)N + (long long)seg * SL, hi = lo + SL;
    const long long bend = (b + 1) * (long long)N;
    if (hi > bend) hi = bend;
    if (lo < row0) lo = row0;                                      // frames of earlier calls are in the carried state
    if (hi > row0 + (long long)ns) hi = row0 + (long long)ns;
    ct_seg r;
    r.len = hi > lo ? (int)(hi - lo) : 0;
    r.lo = lo;
    // position inside the block of the first staged row.  An EMPTY segment (the call does not reach it) keeps the segment's own
    // start: k_ct_goertzel derives its weight-window index from n0 and runs its fixed-trip MFMA loop over the (all-zero) rows
    // regardless -- with n0 = 0 (round 3) that index ran up to e_pos = 2441 floats past the window, through the rest of the
    // workgroup's LDS and beyond its allocation, and 0 x (whatever the previous workgroup left there) is NaN when that is a NaN
    r.n0 = r.len ? (unsigned)(lo - b * (long long)N) : seg * SL;
    return r;
}

// dc blocker of ctcss_execute (:606), pass 1: the zero-state aggregate of every (segment, channel),
//   agg = sum_i lam^(len-1-i) x[lo + i].  A workgroup = one segment x 16 channel slots x 16 time slices of CT_SUB frames: a thread's
// loads are all in flight before its fma chain runs, the 16 slices of a channel are strung together through LDS.
#define CT_SUB 10                                                  /* 16 x 10 >= SL = 153 */
__global__ __launch_bounds__(256) void k_ct_seg_agg(const float *__restrict__ lp, unsigned long long row_mask, long long row0,
                                                    unsigned ns, unsigned M, unsigned N, float lam,
                                                    const float *__restrict__ lampow /*[SL + 1] lam^n*/, float *__restrict__ agg,
                                                    long long b0, const unsigned *__restrict__ chan_list, unsigned n_chan)
{
    __shared__ float sv[16][17];
    __shared__ int sl_len[16];
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;
    const unsigned g = blockIdx.x, blk = g / PMR_CT_SEG, seg = g % PMR_CT_SEG;
    const ct_seg sg = ct_segment(b0 + blk, seg, N, SL, row0, ns);
    const unsigned tid = threadIdx.x, sl = tid & 15u, ss = tid >> 4, ci = blockIdx.y * 16u + sl;
    const unsigned k = ci < n_chan ? (chan_list ? chan_list[ci] : ci) : 0u;
    const int i0 = (int)ss * CT_SUB, len = sg.len - i0 < 0 ? 0 : (sg.len - i0 > CT_SUB ? CT_SUB : sg.len - i0);
    float x[CT_SUB];
#pragma unroll
    for (int u = 0; u < CT_SUB; u++)
        x[u] = (u < len && ci < n_chan) ? lp[((unsigned long long)(sg.lo + i0 + u) & row_mask) * M + k] : 0.f;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < CT_SUB; u++) if (u < len) v = fmaf(lam, v, x[u]);
    sv[ss][sl] = v;
    if (sl == 0) sl_len[ss] = len;
    __syncthreads();
    if (ss == 0 && ci < n_chan) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) a = fmaf(lampow[sl_len[q]], a, sv[q][sl]);
        agg[(size_t)g * M + k] = a;
    }
}

// pass 1 for an OPEN-CHANNEL LIST: the sixteen slots of a workgroup are sixteen (segment, channel) pairs, pair p = segment * n_chan +
// list index (k_ct_seg_agg's sixteen slots are sixteen channels of ONE segment: one open channel ran 2291 workgroups at cfg2 with one
// useful slot each).  Same arithmetic per (segment, channel).
__global__ __launch_bounds__(256) void k_ct_seg_agg_pairs(const float *__restrict__ lp, unsigned long long row_mask, long long row0,
                                                          unsigned ns, unsigned M, unsigned N, float lam,
                                                          const float *__restrict__ lampow, float *__restrict__ agg,
                                                          long long b0, const unsigned *__restrict__ chan_list, unsigned n_chan, unsigned nseg)
{
    __shared__ float sv[16][17];
    __shared__ int sl_len[16][17];
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;
    const unsigned tid = threadIdx.x, sl = tid & 15u, ss = tid >> 4;
    const unsigned long long pair = 16ull * blockIdx.x + sl;
    const unsigned g = (unsigned)(pair / n_chan), ci = (unsigned)(pair - (unsigned long long)g * n_chan);
    const bool on = g < nseg;
    const ct_seg sg = ct_segment(b0 + (on ? g / PMR_CT_SEG : 0u), on ? g % PMR_CT_SEG : 0u, N, SL, row0, ns);
    const unsigned k = on ? chan_list[ci] : 0u;
    const int slen = on ? sg.len : 0;
    const int i0 = (int)ss * CT_SUB, len = slen - i0 < 0 ? 0 : (slen - i0 > CT_SUB ? CT_SUB : slen - i0);
    float x[CT_SUB];
#pragma unroll
    for (int u = 0; u < CT_SUB; u++)
        x[u] = u < len ? lp[((unsigned long long)(sg.lo + i0 + u) & row_mask) * M + k] : 0.f;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < CT_SUB; u++) if (u < len) v = fmaf(lam, v, x[u]);
    sv[ss][sl] = v;
    sl_len[ss][sl] = len;
    __syncthreads();
    if (ss == 0 && on) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) a = fmaf(lampow[sl_len[q][sl]], a, sv[q][sl]);
        agg[(size_t)g * M + k] = a;
    }
}

// pass 2: the blocker's state just before every segment -- a first-order linear scan over the segment aggregates, one WORKGROUP per
// open channel: the aggregates are fetched in one batch, a thread walks a contiguous run of segments, the runs are strung together
// by a Hillis-Steele scan of (multiplier, value) pairs.  Leaves the state after the call's last frame in state[k].
#define CT_PER 24                                                  /* segments per thread: 256 x 24 = 6144 segments = 384 Goertzel blocks per call */
__global__ __launch_bounds__(256) void k_ct_seg_scan(const float *__restrict__ agg, unsigned nseg, unsigned M, unsigned N,
                                                     long long row0, unsigned ns, long long b0,
                                                     const float *__restrict__ lampow, float *__restrict__ state,
                                                     float *__restrict__ W, const unsigned *__restrict__ chan_list)
{
    __shared__ float sP[256], sA[256];
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;
    const unsigned k = chan_list ? chan_list[blockIdx.x] : blockIdx.x, t = threadIdx.x;
    const unsigned per = (nseg + 255u) / 256u, c0 = t * per;       // per <= CT_PER (launcher)
    float a[CT_PER], l[CT_PER];
#pragma unroll
    for (unsigned u = 0; u < CT_PER; u++) {
        const unsigned g = c0 + u;
        const bool in = u < per && g < nseg;
        a[u] = in ? agg[(size_t)g * M + k] : 0.f;
        l[u] = in ? lampow[ct_segment(b0 + g / PMR_CT_SEG, g % PMR_CT_SEG, N, SL, row0, ns).len] : 1.f;
    }
    const float s0 = state[k];                                     // read BEFORE the barriers below: thread 255 overwrites it at the end
    float P = 1.f, A = 0.f;                                        // run from zero state: v_out = P v_in + A
#pragma unroll
    for (unsigned u = 0; u < CT_PER; u++) { A = fmaf(l[u], A, a[u]); P *= l[u]; }
    sP[t] = P; sA[t] = A;
    __syncthreads();
#pragma unroll
    for (unsigned d = 1; d < 256; d <<= 1) {                       // inclusive scan of the affine maps
        float p2 = 1.f, a2 = 0.f;
        if (t >= d) { p2 = sP[t - d]; a2 = sA[t - d]; }
        __syncthreads();
        if (t >= d) { sA[t] = fmaf(sP[t], a2, sA[t]); sP[t] = sP[t] * p2; }
        __syncthreads();
    }
    float v = t == 0 ? s0 : fmaf(sP[t - 1], s0, sA[t - 1]);        // state before this thread's run
#pragma unroll
    for (unsigned u = 0; u < CT_PER; u++) {
        const unsigned g = c0 + u;
        if (u < per && g < nseg) { W[(size_t)g * M + k] = v; v = fmaf(l[u], v, a[u]); }
    }
    if (t == 255) state[k] = fmaf(sP[255], s0, sA[255]);
}

// partial Goertzel sums of one (block, segment, group of 16 channel slots): part[(blk*CT_SEG + seg)][k][j][2].
// A segment is the product  [16 channels] x [frames] . [frames] x [38 tones x (u0, u1)]  with the weights U, k-ordered oldest
// frame first like the recurrence -- a small GEMM, so it runs on the matrix pipe: v_mfma_f32_16x16x4_f32 with
//   A[i][kappa] = x[frame 4 s + kappa][slot i]          (64 consecutive floats of the staged, dc-blocked segment per step)
//   B[kappa][c] = U-weight of frame 4 s + kappa for column c = 2 tone + (0: u0, 1: u1), five 16-column tiles (76 columns used)
// exact f32, sequential in kappa: the sums are those of the scalar loop this replaces.  A workgroup = 5 waves, one column tile
// each (equal MFMA work per wave: with 3 waves sharing 5 tiles 2 : 2 : 1 the busiest SIMD set the kernel's time).
#define CG_T 320
#define CG_DC 192                                                  /* threads of the dc-blocker pass: 12 slices x 16 slots */
#define CG_ROWS 156                                                /* staged rows: SL = 153 rounded up to the MFMA's k step */
#define CG_SLOTS 1024u                                             /* workgroups the chip holds at once (34 KB of LDS, 5 waves each: 4 per CU) */
typedef float ct_f32x4 __attribute__((ext_vector_type(4)));

// A workgroup walks the same segment position of `nb` consecutive blocks (the launcher sizes nb so that the grid is ONE round of
// workgroups): the window of U depends on the position inside the block only and is staged once, and the samples + blocker state
// of block i + 1 are requested BEFORE block i is processed -- per block the workgroup then pays arithmetic, not a chain of L2
// round trips (U window, samples, carried state, one after the other: what made the one-block-per-workgroup form of this
// kernel 39 us whatever the channel count).
#define CG_NG ((CG_ROWS * 16 + CG_T - 1) / CG_T)                    /* staged 4-byte loads per thread (open-channel list) */
#define CG_NX ((CG_ROWS * 4 + CG_T - 1) / CG_T)                     /* staged 16-byte loads per thread (all channels) */
static_assert(CG_T % 16 == 0 && CG_NG <= 16 && 4 * CG_NX <= 16, "ct_xregs holds a thread's share");
struct ct_xregs { float v[16]; float w; };                         // a thread's share of a staged segment + its slot's carried state

// OPEN-CHANNEL LIST (GATHER): slot s of a workgroup is the s-th (block, channel) PAIR of its group, pairs in block-major order
// (pair p = block * n_chan + list index) -- with one open channel (the reference's mode, :893) the sixteen slots are sixteen
// consecutive blocks of that channel, with sixteen open channels one block of all of them.  Round 4 gave every workgroup sixteen
// CHANNEL slots: one open channel ran 768 workgroups of 34 KB and five waves at cfg2, fifteen sixteenths of every MFMA on zeros, and
// cost the chain 8.6 % (profiles/r05_ab_log.txt r5f).  A slot's samples are staged at their position INSIDE the full segment (row r
// <-> block position s_pos + r), so that a slot whose segment is clipped by the call's first frame shares the weight window of the
// others: rows in front of the clipped segment's first frame are zero.
struct ct_slot { long long blk; unsigned k; bool on; };
static __device__ __forceinline__ ct_slot ct_pair_slot(unsigned long long pair, unsigned nblk, unsigned n_chan,
                                                       const unsigned *__restrict__ chan_list, long long b0)
{
    ct_slot sl;
    const unsigned long long blk = pair / n_chan;
    sl.on = blk < nblk;
    sl.blk = b0 + (long long)(sl.on ? blk : 0ull);
    sl.k = sl.on ? chan_list[(unsigned)(pair - blk * n_chan)] : 0u;
    return sl;
}

template <bool GATHER>
static __device__ __forceinline__ void ct_load_x(ct_xregs &x, const float *__restrict__ lp, unsigned long long row_mask, unsigned M,
                                                 const ct_seg &sg, unsigned cg, unsigned n_chan, unsigned kch, unsigned tid,
                                                 const float *__restrict__ W, unsigned gseg, int roff = 0, bool slot_on = true)
{
    if constexpr (GATHER) {                                        // the thread's own slot: sg / roff / kch / gseg are the SLOT's
#pragma unroll
        for (int u = 0; u < CG_NG; u++) {                          // CG_NG x CG_T >= CG_ROWS x 16 (CG_T is a multiple of 16: one slot per thread)
            const int r = (int)((tid + CG_T * u) >> 4) - roff;     // frame sg.lo + r sits in staged row r + roff
            x.v[u] = (r >= 0 && r < sg.len && slot_on) ? lp[((unsigned long long)(sg.lo + r) & row_mask) * M + kch] : 0.f;
        }
        x.w = slot_on ? W[(size_t)gseg * M + kch] : 0.f;
        return;
    } else {
#pragma unroll
        for (int u = 0; u < CG_NX; u++) {                          // samples: 64 bytes per frame
            const int i = (int)tid + CG_T * u, r = i >> 2, q4 = (i & 3) * 4;
            const float *src = lp + ((unsigned long long)(sg.lo + r) & row_mask) * M + cg + q4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < sg.len && i < CG_ROWS * 4) {
                if (cg + q4 + 4 <= M) t = *reinterpret_cast<const float4 *>(src);         // M >= 4 is a power of two
                else if (cg + q4 < M) { t.x = src[0]; if (cg + q4 + 1 < M) t.y = src[1]; }   // M = 2
            }
            x.v[4 * u] = t.x; x.v[4 * u + 1] = t.y; x.v[4 * u + 2] = t.z; x.v[4 * u + 3] = t.w;
        }
    }
    x.w = cg + (tid & 15u) < n_chan ? W[(size_t)gseg * M + kch] : 0.f;
}

template <bool GATHER>
static __device__ __forceinline__ void ct_store_x(const ct_xregs &x, float *xs, unsigned tid)
{
    if constexpr (GATHER) {
#pragma unroll
        for (int u = 0; u < CG_NG; u++) if (tid + CG_T * u < CG_ROWS * 16) xs[tid + CG_T * u] = x.v[u];
    } else {
#pragma unroll
        for (int u = 0; u < CG_NX; u++) {
            const int i = (int)tid + CG_T * u;
            if (i < CG_ROWS * 4) *reinterpret_cast<float4 *>(xs + 4 * i) = make_float4(x.v[4 * u], x.v[4 * u + 1], x.v[4 * u + 2], x.v[4 * u + 3]);
        }
    }
}

// (every channel open: slot = channel cg + s of ONE block; an open-channel list runs k_ct_goertzel_pairs below)
__global__ __launch_bounds__(CG_T) void k_ct_goertzel(const float *__restrict__ lp, unsigned long long row_mask,
                                                     long long row0, unsigned ns, unsigned M, unsigned N,
                                                     const float *__restrict__ U /*[38][N+1], U[j][m+1] = U_m*/,
                                                     float *__restrict__ part, long long b0, unsigned nblk, unsigned nb,
                                                     const unsigned *__restrict__ chan_list, unsigned n_chan,
                                                     const float *__restrict__ W /*[segments][M] blocker state before each segment*/,
                                                     float dc_a1, const float *__restrict__ lampow)
{
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;         // frames per segment (<= 153)
    float *xs = reinterpret_cast<float *>(smem_c);                 // [CG_ROWS][16], rows beyond the segment zero
    float *us = xs + (size_t)CG_ROWS * 16;                         // [38][US]: us[j][m] = U[j][N - e + m], e = segment end inside the block
    const unsigned US = SL + 2 + ((SL & 1) ? 0 : 1);               // row stride of us, odd
    float *sagg = us + (size_t)PMR_CT_TONES * US;                  // [12][16]
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned seg = blockIdx.x % PMR_CT_SEG, blk0 = (blockIdx.x / PMR_CT_SEG) * nb, cg = blockIdx.y * 16u;   // cg: first of 16 channel SLOTS
    const unsigned nbw = blk0 + nb <= nblk ? nb : nblk - blk0;     // blocks of this workgroup (>= 1: launcher)
    const unsigned e_pos = min(N, (seg + 1) * SL), s_pos = seg * SL;     // the FULL segment inside a block: [s_pos, e_pos)
    __shared__ float s_lam[16];                                    // lambda^n, n <= 13 (a table read inside the serial carry loop below
    if (tid < 16) s_lam[tid] = lampow[tid];                        //  must not be a global load per step)
    const unsigned slot = cg + (tid & 15u);
    const unsigned kch = slot < n_chan ? slot : 0u;                // channel of this thread's slot (staging, carry)
    // ---- U window of the full segment, once: weight of block position p is us[j][e_pos - p] (-> u0) and us[j][e_pos - p - 1] (-> u1);
    //      and the first block's samples: every load of the thread is in flight before the first is used ----
    ct_xregs xq;
    ct_seg sg = ct_segment(b0 + blk0, seg, N, SL, row0, ns);
    {
        constexpr int NJ = (PMR_CT_TONES + CG_T / 64 - 1) / (CG_T / 64), NM = (CG_ROWS + 1 + 63) / 64;
        float uv[NJ][NM];
        const unsigned wlen = e_pos - s_pos;
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
            const unsigned j = wave + (CG_T / 64) * jj;
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
                const unsigned m = lane + 64u * mm;
                uv[jj][mm] = (j < PMR_CT_TONES && m <= wlen) ? U[(size_t)j * (N + 1) + (N - e_pos) + m] : 0.f;
            }
        }
        ct_load_x<false>(xq, lp, row_mask, M, sg, cg, n_chan, kch, tid, W, blk0 * PMR_CT_SEG + seg);
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
            const unsigned j = wave + (CG_T / 64) * jj;
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
                const unsigned m = lane + 64u * mm;
                if (j < PMR_CT_TONES && m < US) us[j * US + m] = uv[jj][mm];      // the row's tail (m > wlen) is zero
            }
        }
    }
    ct_store_x<false>(xq, xs, tid);
    float wv = xq.w;
    __syncthreads();

    for (unsigned bi = 0; bi < nbw; bi++) {
        const unsigned blk = blk0 + bi, gseg = blk * PMR_CT_SEG + seg;
        const int len = sg.len;
        const unsigned n0 = sg.n0;
        const bool more = bi + 1 < nbw;
        if (more) {                                                // the next block's samples and state travel while this one is processed
            sg = ct_segment(b0 + blk + 1, seg, N, SL, row0, ns);
            ct_load_x<false>(xq, lp, row_mask, M, sg, cg, n_chan, kch, tid, W, gseg + PMR_CT_SEG);
        }
        // ---- dc blocker of ctcss_execute (:606), pass 3, on the staged samples in place: thread (slot, slice of CT_SUBG frames)
        // strings the zero-state aggregates of the slices before its own onto the segment's carried state W, then runs the exact
        // recurrence v0 = x - a1 v1, y = v0 - v1 (liquid's iirfilt_rrrf, individually rounded) over its slice ----
        {
            constexpr int CT_SUBG = 13;                            // 12 slices x 13 >= SL = 153: the first CG_DC = 12 x 16 threads
            const unsigned sl = tid & 15u, ss = tid >> 4;
            const bool dcw = tid < CG_DC;
            const int i0 = (int)ss * CT_SUBG, ln = !dcw ? 0 : len - i0 < 0 ? 0 : (len - i0 > CT_SUBG ? CT_SUBG : len - i0);
            const float lam = -dc_a1;
            float x[CT_SUBG];
#pragma unroll
            for (int u = 0; u < CT_SUBG; u++) x[u] = u < ln ? xs[(i0 + u) * 16 + sl] : 0.f;
            float v = 0.f;
#pragma unroll
            for (int u = 0; u < CT_SUBG; u++) if (u < ln) v = fmaf(lam, v, x[u]);
            if (dcw) sagg[ss * 16 + sl] = v;
            __syncthreads();
            float v1 = wv;
            for (unsigned q = 0; q < (dcw ? ss : 0u); q++) {
                const int lq = len - (int)q * CT_SUBG;
                v1 = fmaf(s_lam[lq < 0 ? 0 : (lq > CT_SUBG ? CT_SUBG : lq)], v1, sagg[q * 16 + sl]);
            }
#pragma unroll
            for (int u = 0; u < CT_SUBG; u++) {
                if (u < ln) {
                    const float v0 = __fsub_rn(x[u], __fmul_rn(dc_a1, v1));
                    xs[(i0 + u) * 16 + sl] = __fsub_rn(v0, v1);
                    v1 = v0;
                }
            }
            __syncthreads();
        }
        // ---- the Goertzel sums on the matrix pipe.  Row r of the staged segment sits at block position p = n0 + r ----
        {
            const int col = lane & 15, kk = lane >> 4;
            const int d0 = (int)e_pos - (int)n0;                   // us index of row r, column (tone, which): d0 - r - which
            ct_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // Every row of the staged segment beyond `len` is zero, so all CG_ROWS / 4 k steps always run (a zero sample times a finite
            // weight adds nothing): a fixed trip count, fully unrolled, every LDS address = base + immediate, the reads of 13 steps
            // issued ahead of their MFMAs.  (With a run-time trip count and a clamped weight index the compiler issued read, wait,
            // MFMA one after the other: ~400 cycles per step, 21 of this kernel's 33 us.)  Weight index of step st: ui - 4 st, which
            // runs below the row's start for the zero rows of a clipped segment -- into the previous tone's row or the sample area:
            // finite numbers (the rows' tails are zero-filled above, every staged sample row is written), multiplied by zero.
            // Bounds: n0 in [s_pos, e_pos] (ct_segment, empty segments included) => d0 in [0, SL]; the highest index read is
            // jt US + d0 <= 37 US + SL < 38 US (inside `us`), the lowest jt US + d0 - 1 - (CG_ROWS - 1) >= -CG_ROWS (inside `xs`,
            // which lies directly below `us`): never outside what this workgroup wrote.  tests run under the LDS poison mode.
            constexpr int NST = CG_ROWS / 4;
            const int c = 16 * (int)wave + col, jt = c >> 1;       // column -> (tone, which); columns >= 76: tone 37 again, never stored
            const float *pb = us + (size_t)(jt < (int)PMR_CT_TONES ? jt : (int)PMR_CT_TONES - 1) * US + (d0 - (c & 1) - kk) - 4 * (NST - 1);
            const float *xa = xs + lane;                           // step s: xa[64 s]
            constexpr int CH = 13;                                 // k steps per chunk: the chunk's 26 LDS reads are issued together,
            static_assert(NST % CH == 0, "chunks cover the steps");//  then its 13 MFMAs run (scheduling barriers keep it that way)
#pragma unroll
            for (int s0 = 0; s0 < NST; s0 += CH) {
                float av[CH], bv[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    av[u] = xa[64 * (s0 + u)];
                    bv[u] = pb[4 * (NST - 1 - s0 - u)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // D: lane holds column `col` of its tile, register q = channel slot 4 kk + q
            if (c < 2 * (int)PMR_CT_TONES) {
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const unsigned sl = cg + 4 * kk + qq;
                    if (sl < n_chan) {
                        const unsigned k = sl;
                        part[((size_t)gseg * M + k) * PMR_CT_TONES * 2 + c] = acc[qq];
                    }
                }
            }
        }
        if (more) {
            __syncthreads();                                       // every reader of the staged segment is done
            ct_store_x<false>(xq, xs, tid);
            wv = xq.w;
            __syncthreads();
        }
    }
}

// The same kernel for an OPEN-CHANNEL LIST: the sixteen slots of a workgroup are (block, channel) pairs (ct_pair_slot) at one segment
// position; a workgroup walks `nb` consecutive groups of sixteen pairs.  Row r of a slot holds block position s_pos + r, whichever
// frame the call reaches the segment at: the weight window is the full segment's for every slot (d0 = e_pos - s_pos), the slot's
// dc-blocker pass starts at its own first row `roff` from the state carried into the segment.
__global__ __launch_bounds__(CG_T) void k_ct_goertzel_pairs(const float *__restrict__ lp, unsigned long long row_mask,
                                                           long long row0, unsigned ns, unsigned M, unsigned N,
                                                           const float *__restrict__ U, float *__restrict__ part, long long b0,
                                                           unsigned nblk, unsigned nb, unsigned ngroups,
                                                           const unsigned *__restrict__ chan_list, unsigned n_chan,
                                                           const float *__restrict__ W, float dc_a1, const float *__restrict__ lampow)
{
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG;
    float *xs = reinterpret_cast<float *>(smem_c);                 // [CG_ROWS][16]
    float *us = xs + (size_t)CG_ROWS * 16;                         // [38][US]
    const unsigned US = SL + 2 + ((SL & 1) ? 0 : 1);
    float *sagg = us + (size_t)PMR_CT_TONES * US;                  // [12][16]
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned seg = blockIdx.x % PMR_CT_SEG, g0 = (blockIdx.x / PMR_CT_SEG) * nb;
    const unsigned nbw = g0 + nb <= ngroups ? nb : ngroups - g0;   // groups of this workgroup (>= 1: launcher)
    const unsigned e_pos = min(N, (seg + 1) * SL), s_pos = seg * SL, wlen = e_pos - s_pos;
    __shared__ float s_lam[16];
    if (tid < 16) s_lam[tid] = lampow[tid];
    const unsigned myslot = tid & 15u;
    // a thread's slot of group g: segment clipped to the call, its first staged row, its (segment, channel) index
    struct slotv { ct_seg sg; int roff; unsigned k, gseg; bool on; };
    const auto slot_of = [&](unsigned g, unsigned sidx) {
        slotv v;
        const ct_slot sl = ct_pair_slot(16ull * g + sidx, nblk, n_chan, chan_list, b0);
        v.sg = ct_segment(sl.blk, seg, N, SL, row0, ns);
        if (!sl.on) v.sg.len = 0;
        v.roff = v.sg.len ? (int)(v.sg.lo - (sl.blk * (long long)N + (long long)s_pos)) : 0;
        v.k = sl.k; v.gseg = (unsigned)(sl.blk - b0) * PMR_CT_SEG + seg; v.on = sl.on;
        return v;
    };
    ct_xregs xq;
    slotv cur = slot_of(g0, myslot);
    {
        constexpr int NJ = (PMR_CT_TONES + CG_T / 64 - 1) / (CG_T / 64), NM = (CG_ROWS + 1 + 63) / 64;
        float uv[NJ][NM];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
            const unsigned j = wave + (CG_T / 64) * jj;
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
                const unsigned m = lane + 64u * mm;
                uv[jj][mm] = (j < PMR_CT_TONES && m <= wlen) ? U[(size_t)j * (N + 1) + (N - e_pos) + m] : 0.f;
            }
        }
        ct_load_x<true>(xq, lp, row_mask, M, cur.sg, 0u, n_chan, cur.k, tid, W, cur.gseg, cur.roff, cur.on);
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) {
            const unsigned j = wave + (CG_T / 64) * jj;
#pragma unroll
            for (int mm = 0; mm < NM; mm++) {
                const unsigned m = lane + 64u * mm;
                if (j < PMR_CT_TONES && m < US) us[j * US + m] = uv[jj][mm];      // the row's tail (m > wlen) is zero
            }
        }
    }
    ct_store_x<true>(xq, xs, tid);
    float wv = xq.w;
    __syncthreads();

    for (unsigned gi = 0; gi < nbw; gi++) {
        const unsigned g = g0 + gi;
        const int len = cur.sg.len, roff = cur.roff;
        const bool more = gi + 1 < nbw;
        if (more) {                                                // the next group's samples and state travel while this one is processed
            cur = slot_of(g + 1, myslot);
            ct_load_x<true>(xq, lp, row_mask, M, cur.sg, 0u, n_chan, cur.k, tid, W, cur.gseg, cur.roff, cur.on);
        }
        // ---- dc blocker (:606) on the staged samples in place, per slot from row `roff`: k_ct_goertzel's pass 3 ----
        {
            constexpr int CT_SUBG = 13;
            const unsigned sl = myslot, ss = tid >> 4;
            const bool dcw = tid < CG_DC;
            const int i0 = (int)ss * CT_SUBG, ln = !dcw ? 0 : len - i0 < 0 ? 0 : (len - i0 > CT_SUBG ? CT_SUBG : len - i0);
            const float lam = -dc_a1;
            float x[CT_SUBG];
#pragma unroll
            for (int u = 0; u < CT_SUBG; u++) x[u] = u < ln ? xs[(roff + i0 + u) * 16 + sl] : 0.f;
            float v = 0.f;
#pragma unroll
            for (int u = 0; u < CT_SUBG; u++) if (u < ln) v = fmaf(lam, v, x[u]);
            if (dcw) sagg[ss * 16 + sl] = v;
            __syncthreads();
            float v1 = wv;
            for (unsigned q = 0; q < (dcw ? ss : 0u); q++) {
                const int lq = len - (int)q * CT_SUBG;
                v1 = fmaf(s_lam[lq < 0 ? 0 : (lq > CT_SUBG ? CT_SUBG : lq)], v1, sagg[q * 16 + sl]);
            }
#pragma unroll
            for (int u = 0; u < CT_SUBG; u++) {
                if (u < ln) {
                    const float v0 = __fsub_rn(x[u], __fmul_rn(dc_a1, v1));
                    xs[(roff + i0 + u) * 16 + sl] = __fsub_rn(v0, v1);
                    v1 = v0;
                }
            }
            __syncthreads();
        }
        // ---- the Goertzel sums on the matrix pipe: row r <-> block position s_pos + r, weight index wlen - r - which ----
        {
            const int col = lane & 15, kk = lane >> 4;
            const int d0 = (int)wlen;
            ct_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // fixed trip count over all CG_ROWS rows (rows outside a slot's frames are zero).  Bounds as in k_ct_goertzel with
            // d0 = wlen in [1, SL]: highest index jt US + wlen < 38 US, lowest jt US + wlen - 1 - (CG_ROWS - 1) >= -CG_ROWS (inside xs)
            constexpr int NST = CG_ROWS / 4;
            const int c = 16 * (int)wave + col, jt = c >> 1;
            const float *pb = us + (size_t)(jt < (int)PMR_CT_TONES ? jt : (int)PMR_CT_TONES - 1) * US + (d0 - (c & 1) - kk) - 4 * (NST - 1);
            const float *xa = xs + lane;
            constexpr int CH = 13;
            static_assert(NST % CH == 0, "chunks cover the steps");
#pragma unroll
            for (int s0 = 0; s0 < NST; s0 += CH) {
                float av[CH], bv[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    av[u] = xa[64 * (s0 + u)];
                    bv[u] = pb[4 * (NST - 1 - s0 - u)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c < 2 * (int)PMR_CT_TONES) {
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const ct_slot so = ct_pair_slot(16ull * g + 4u * (unsigned)kk + (unsigned)qq, nblk, n_chan, chan_list, b0);
                    if (so.on) {
                        const unsigned gs = (unsigned)(so.blk - b0) * PMR_CT_SEG + seg;
                        part[((size_t)gs * M + so.k) * PMR_CT_TONES * 2 + c] = acc[qq];
                    }
                }
            }
        }
        if (more) {
            __syncthreads();
            ct_store_x<true>(xq, xs, tid);
            wv = xq.w;
            __syncthreads();
        }
    }
}

// reduce the segments (fixed order), add the carry of a block begun in an earlier call, decide (:381-406).
// One 64-thread workgroup per (block, channel): lane j < 38 sums tone j over the segments, lane 0 takes the decision.
__global__ __launch_bounds__(64) void k_ct_final(const float *__restrict__ part, unsigned nblk, unsigned ncomplete,
                                                 unsigned M, const float *__restrict__ coef,
                                                 const float *__restrict__ carry_in, float *__restrict__ carry_out,
                                                 pmr_ctcss_event *__restrict__ events, unsigned char *__restrict__ restart,
                                                 const unsigned *__restrict__ chan_list, unsigned n_chan)
{
    __shared__ float spw[PMR_CT_TONES];
    const unsigned ci = blockIdx.x % n_chan, blk = blockIdx.x / n_chan, j = threadIdx.x;
    const unsigned k = chan_list ? chan_list[ci] : ci;
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
        if (blk + 1 == nblk) {                                     // partial sums of the block in progress (none: zeros) for the next call
            carry_out[((size_t)k * PMR_CT_TONES + j) * 2] = complete ? 0.f : u0;
            carry_out[((size_t)k * PMR_CT_TONES + j) * 2 + 1] = complete ? 0.f : u1;
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
        // the block in progress when the channel's detector was restarted (ctcss_detector_reset, :867 -- pmr_chain.c
        // ct_restart_channel) holds only the frames since: no decision.  (blk == 0: the one block that carries sums of earlier calls)
        if (blk == 0 && restart[k]) { e.index = -1; e.detected = 0; e.max_power = 0.f; e.avg_power = 0.f; restart[k] = 0; }
        events[(size_t)blk * M + k] = e;
    }
}

extern "C" unsigned pmr_ct_max_segments(void) { return 256u * CT_PER; }

extern "C" int pmr_launch_ct_detector(pmr_stream_t s, const float *lp, uint64_t row_mask, int64_t row0, unsigned ns,
                                      unsigned M, unsigned N, float a1, const float *lampow, float *state, float *agg, float *W,
                                      const float *U, const float *coef, float *part,
                                      const float *carry_in, float *carry_out, pmr_ctcss_event *events, unsigned char *restart,
                                      unsigned nblk, unsigned ncomplete, const unsigned *chan_list, unsigned n_chan)
{
    const unsigned nc = chan_list ? n_chan : M;
    if (!ns || !nblk || !nc) return 0;
    const long long b0 = row0 / (long long)N;
    const unsigned nseg = nblk * PMR_CT_SEG;
    if (nseg > pmr_ct_max_segments()) return (int)hipErrorInvalidValue;        /* (pmr_chain_ctcss_enable checks the plan up front) */
    const unsigned SL = (N + PMR_CT_SEG - 1) / PMR_CT_SEG, US = SL + 2 + ((SL & 1) ? 0 : 1);
    if (SL > 16 * CT_SUB || SL > 12 * 13 || SL > CG_ROWS) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)s;
    const unsigned long long rm = (unsigned long long)row_mask;
    /* (EXP_CT_NO_*: timing experiments of tools/ab_libs.py --ctcss -- the detector without one of its kernels.  WRONG results) */
#ifndef EXP_CT_NO_AGG
    if (chan_list)
        PMR_KLAUNCH(k_ct_seg_agg_pairs, dim3((unsigned)(((unsigned long long)nseg * nc + 15ull) / 16ull)), dim3(256), 0, st, lp, rm, (long long)row0,
                    ns, M, N, -a1, lampow, agg, b0, chan_list, nc, nseg);
    else
        PMR_KLAUNCH(k_ct_seg_agg, dim3(nseg, (nc + 15) / 16), dim3(256), 0, st, lp, rm, (long long)row0, ns, M, N, -a1, lampow, agg,
                       b0, chan_list, nc);
#endif
#ifndef EXP_CT_NO_SCAN
    PMR_KLAUNCH(k_ct_seg_scan, dim3(nc), dim3(256), 0, st, agg, nseg, M, N, (long long)row0, ns, b0, lampow, state, W, chan_list);
#endif
#ifndef EXP_CG_EXTRA_LDS
#define EXP_CG_EXTRA_LDS 0      /* sensitivity experiment: bytes of unused LDS per Goertzel workgroup */
#endif
    const size_t lds = ((size_t)CG_ROWS * 16 + (size_t)PMR_CT_TONES * US + 12 * 16) * sizeof(float) + EXP_CG_EXTRA_LDS;
#ifndef EXP_CT_NO_GOERTZEL
    if (chan_list) {
        /* open-channel list: slots are (block, channel) pairs, sixteen per group; the grid is one round of resident workgroups */
        const unsigned long long pairs = (unsigned long long)nblk * nc;
        const unsigned ngroups = (unsigned)((pairs + 15ull) / 16ull);
        unsigned nbp = (unsigned)(((unsigned long long)ngroups * PMR_CT_SEG + CG_SLOTS - 1) / CG_SLOTS);
        if (nbp < 1) nbp = 1;
        PMR_KLAUNCH(k_ct_goertzel_pairs, dim3(((ngroups + nbp - 1) / nbp) * PMR_CT_SEG), dim3(CG_T), lds, st, lp, rm, (long long)row0, ns, M, N, U,
                    part, b0, nblk, nbp, ngroups, chan_list, nc, W, a1, lampow);
    } else {
        /* every channel open: sixteen channel slots of one block per workgroup pass; blocks per workgroup: one round of resident workgroups */
        const unsigned ny = (nc + 15) / 16;
        unsigned nb = (unsigned)(((unsigned long long)nblk * PMR_CT_SEG * ny + CG_SLOTS - 1) / CG_SLOTS);
        if (nb < 1) nb = 1;
        const dim3 grid(((nblk + nb - 1) / nb) * PMR_CT_SEG, ny);
        PMR_KLAUNCH(k_ct_goertzel, grid, dim3(CG_T), lds, st, lp, rm, (long long)row0, ns, M, N, U, part, b0, nblk, nb, chan_list, nc,
                           W, a1, lampow);
    }
#endif
#ifndef EXP_CT_NO_FINAL
    PMR_KLAUNCH(k_ct_final, dim3(nblk * nc), dim3(64), 0, st, part, nblk, ncomplete, M, coef, carry_in, carry_out, events,
                       restart, chan_list, nc);
#endif
    return (int)hipGetLastError();
}
