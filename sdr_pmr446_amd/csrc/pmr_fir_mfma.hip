// pmr_fir_mfma.hip -- the audio FIR (reference src/sdr_pmr446.c:882-904: 377-tap CTCSS high-pass, gain, 50 us
// de-emphasis, PCM hand-off) on the gfx950 MATRIX pipe, 16 channels per workgroup (M = 16: the whole row; larger M:
// blockIdx.y walks the groups of 16 channels).
//
// Why MFMA here although the chain is "streaming DSP": with all 16 channels demodulated this stage is the FLOP
// hot spot at small decimation ratios (383 MACs per audio sample = 32 MACs per raw input sample at cfg2, 2/3 of all
// arithmetic) and it is compute- not HBM-bound.  Measured on MI355X (tools/ubench/valu_rate.hip): v_fma_f32 peaks at
// ~67 TFLOP/s, v_pk_fma_f32 at ~115 TFLOP/s, f32 MFMA at ~155 TFLOP/s -- and the matrix pipe is otherwise idle while
// the front end and the channelizer saturate the VALU on the other stream.  The f32 MFMA is an exact, k-ordered fmaf
// chain (MI355X_MICROARCH.md), so the accumulation order is the oracle's: oldest sample first.
//
// Formulation.  With gain and the (truncated, 7-term) de-emphasis response folded into the taps g[0..n) on the host,
//   Y[t][ch] = sum_d g[d] X[t-d][ch]
// is a banded Toeplitz matrix times the data:  D[i][j] = sum_kappa A[i][kappa] B[kappa][j] with
//   A[i][kappa] = g[i + (n-1) - kappa]   (zero outside the band; read from the zero-padded tap table)
//   B[kappa][j] = X[T_j - (n-1) + kappa][ch_j],      D[i][j] = Y[T_j + i][ch_j]
// A 32x32 output tile covers 32 frames x (16 channels x 2 time blocks).  v_mfma_f32_32x32x2_f32 consumes two kappa
// per instruction: lane l supplies A[l&31][l>>5] and B[l>>5][l&31], i.e. ONE tap and ONE sample per lane per step,
// both from LDS (the workgroup stages its 16-channel slab of the time-major discriminator stream once).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pmr_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FM_NT 256
#define FM_TILE 256                              /* frames per workgroup: 4 waves x 2 blocks x 32 */

static __device__ __forceinline__ int16_t pcm16(float y)
{
    // branch-free (four of these sit in every epilogue store): NaN -> 0, saturation by clamping, truncation toward zero by the cast
    float s = y * 32767.0f;
    s = s == s ? s : 0.f;
    s = __builtin_fminf(__builtin_fmaxf(s, -32768.0f), 32767.0f);
    return (int16_t)(int)s;                       // truncation toward zero (src/dsd_in.c:174), saturated
}

#define FM_PRE 11                                /* float4 prefetch registers per thread: windows up to 704 rows */

// GATHER (open-channel mask, reference src/sdr_pmr446.c:876-877: only the squelch-selected channels are demodulated): the 16
// columns of a tile are 16 arbitrary (channel, 256-frame segment) UNITS instead of 16 adjacent channels at one time -- unit u
// = (enabled channel u / nseg, segment u % nseg) -- so one enabled channel costs 1/16 of a tile row, not a whole tile.  Only the
// window staging (4-byte gathers instead of 64-byte rows) and the store addresses differ; the MFMA loop is the same.
template <bool GLB, int TPW /*tiles per workgroup: 2 = the second tile's window is prefetched under the first tile's MFMAs*/,
          bool SWAP /*operands exchanged: the accumulators hold the TRANSPOSED tile (lane = frame), see the epilogue*/,
          bool GATHER = false,
          bool DUAL = false /*a SECOND tap set over the same samples, written time-major to out2_tm: the CTCSS low-pass branch
                              delay188(x) - hp(x) (reference :884-889) in the same pass as the audio filter -- one window
                              staging and one B operand for two banded-Toeplitz products*/>
__global__ __launch_bounds__(FM_NT, 3) void k_fir_mfma16(const float *__restrict__ in, unsigned long long row_mask,
                                                      long long row0, unsigned ns, const float *__restrict__ taps_c,
                                                      unsigned ntaps, float *__restrict__ out_tm,
                                                      int16_t *__restrict__ pcm, float *__restrict__ audio,
                                                      unsigned stride, unsigned M /*row width, multiple of 16*/,
                                                      const unsigned *__restrict__ chan_list, unsigned n_units, unsigned nseg,
                                                      const float *__restrict__ taps2_c, float *__restrict__ out2_tm)
{
    static_assert(!GATHER || (TPW == 1 && !GLB), "gathered units: one tile per workgroup, LDS window");
    static_assert(!DUAL || (TPW == 1 && !GLB && SWAP), "dual tap sets: one tile per workgroup (register budget), LDS window");
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if constexpr (!GATHER) {
        const unsigned L = pmr_xcd_contiguous(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        bx = L % gridDim.x; by = L / gridDim.x;
    }
    __shared__ unsigned s_ch[16];                                    // channel of column slot s (relative to the pointers below)
    __shared__ long s_t0[16];                                        // first frame of slot s
    if constexpr (GATHER) {
        if (threadIdx.x < 16) {
            const unsigned u = blockIdx.x * 16u + threadIdx.x;
            s_ch[threadIdx.x] = chan_list[u < n_units ? u / nseg : 0];
            s_t0[threadIdx.x] = u < n_units ? (long)(u % nseg) * FM_TILE : (long)ns;      // beyond the block: nothing stored
        }
    } else {
        // a group of 16 channels per workgroup row: the tile is 16 channels wide whatever M is.  (bx, by) = XCD-contiguous
        // re-numbering of the launch's workgroups: consecutive time tiles of one channel group, whose windows overlap, share an L2
        const unsigned cg0 = by * 16u;
        in += cg0;
        if (out_tm) out_tm += cg0;
        if (DUAL) out2_tm += cg0;
        if (pcm) pcm += (size_t)cg0 * stride;
        if (audio) audio += (size_t)cg0 * stride;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_m[];
    float *Qs = reinterpret_cast<float *>(smem_m);                   // [ntaps + 2*PMR_TAP_PAD] padded taps
    const unsigned qlen = ntaps + 2 * PMR_TAP_PAD;
    float *Q2 = Qs + ((qlen + 31) & ~31u);                           // DUAL: second padded tap table
    float *Xs = Q2 + (DUAL ? ((qlen + 31) & ~31u) : 0u);             // rows: r at r*16 + 16*(r>>5)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned nrows = FM_TILE + ntaps + 31;                     // frames T0-(ntaps-1) .. T0+255 (+32: kappa padded to 32)
    const int j = lane & 31, kk = lane >> 5, ch = j & 15, blk = j >> 4;
    const int Tj = 64 * wave + 32 * blk;                             // frame offset of this column inside the tile

    for (unsigned i = tid; i < qlen; i += FM_NT) Qs[i] = taps_c[i];
    if constexpr (DUAL) for (unsigned i = tid; i < qlen; i += FM_NT) Q2[i] = taps2_c[i];

    // window of the tile starting at frame T0: global -> registers (issue), registers -> LDS (commit)
    float4 pre[FM_PRE];
    auto issue = [&](long T0) {
#pragma unroll
        for (int i = 0; i < FM_PRE; i++) {
            const unsigned u = tid + FM_NT * i, r = u >> 2, q4 = (u & 3) * 4;
            const long t = T0 - (long)(ntaps - 1) + r;
            pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < nrows * 4 && t < (long)ns)
                pre[i] = *reinterpret_cast<const float4 *>(in + ((unsigned long long)(row0 + t) & row_mask) * M + q4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < FM_PRE; i++) {
            const unsigned u = tid + FM_NT * i, r = u >> 2, q4 = (u & 3) * 4;
            if (u < nrows * 4) *reinterpret_cast<float4 *>(Xs + r * 16 + 16 * (r >> 5) + q4) = pre[i];
        }
    };
    // GATHER staging: element (row r, slot s) = in[frame s_t0[s] - (ntaps-1) + r][s_ch[s]]: 4-byte loads, 42 per thread
    constexpr int FM_PRE_G = GATHER ? 4 * FM_PRE : 1;
    float preg[FM_PRE_G];
    auto issue_g = [&]() {
#pragma unroll
        for (int i = 0; i < FM_PRE_G; i++) {
            const unsigned e = tid + FM_NT * i, r = e >> 4, sl = e & 15;
            const long t = s_t0[sl] - (long)(ntaps - 1) + r;
            preg[i] = 0.f;
            if (e < nrows * 16 && t < (long)ns) preg[i] = in[((unsigned long long)(row0 + t) & row_mask) * M + s_ch[sl]];
        }
    };
    auto commit_g = [&]() {
#pragma unroll
        for (int i = 0; i < FM_PRE_G; i++) {
            const unsigned e = tid + FM_NT * i, r = e >> 4, sl = e & 15;
            if (e < nrows * 16) Xs[r * 16 + 16 * (r >> 5) + sl] = preg[i];
        }
    };
    // GLB: no sample window in LDS at all -- the B operand comes straight from the time-major ring through the vector L1
    // (a wave-instruction touches four 64-byte rows).  With ~2 KB of LDS and < 128 registers a workgroup of this kernel
    // fits NEXT TO the front end's tiles on a CU.
    const long tile0 = (long)bx * TPW;
    if constexpr (GATHER) { __syncthreads(); issue_g(); commit_g(); }
    else if constexpr (!GLB) { issue(tile0 * FM_TILE); commit(); }
    __syncthreads();

#pragma unroll
    for (int it = 0; it < TPW; it++) {
    const long T0 = GATHER ? 0 : (tile0 + it) * FM_TILE;             // first frame of this tile (relative to row0)
    if (T0 >= (long)ns) break;                                       // uniform
    const bool more = !GLB && !GATHER && it + 1 < TPW && T0 + FM_TILE < (long)ns;
    if (more) issue(T0 + FM_TILE);                                   // in flight during this tile's MFMAs
    const bool active = GATHER || T0 + 64 * wave < (long)ns;         // else: whole wave beyond the block

    if (active) {
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[i] = 0.f; acc2[i] = 0.f; }
    // kappa runs over [0, ntaps + 31), padded up to a multiple of 32 (the extra taps are zeros of the padded table): 16
    // MFMA steps per group.  Inside a group every LDS address is one per-lane base plus a compile-time offset:
    //   tap     for kappa = 2s + kk :  Qs[PAD + (ntaps-1) + (lane&31) - kk - 2s]
    //   sample  for kappa = 2s + kk :  row r = Tj + kk + 2s at r*16 + 16*(r>>5) + ch, and (kk + 2s) >> 5 == s >> 4,
    //                                  so a group of 16 steps advances the base by 16*32 + 16 floats.
    // Explicit two-set software pipeline (no register copies): the 32 operands of group g+1 are in flight from LDS while
    // the 16 MFMAs of group g issue back to back; each set is waited for only right before its first use.
    const unsigned groups = (ntaps + 31 + 31) / 32;
    const float *q0 = Qs + PMR_TAP_PAD + (ntaps - 1) + (lane & 31) - kk - 30;      // group 0, step 15; step s at q[2*(15-s)]
    const float *x0 = Xs + (Tj + kk) * 16 + 16 * ((Tj + kk) >> 5) + ch;            // group 0, step 0;  step s at x[32*s]
    const long long rb0 = row0 + T0 + Tj + kk - (long long)(ntaps - 1);            // GLB: ring row of group 0, step 0
    const float *gin = in + ch;
    float a0[16], b0[16], a1[16], b1[16];
    float c0[DUAL ? 16 : 1], c1[DUAL ? 16 : 1];                                     // DUAL: taps of the second set
    const float *q20 = Q2 + (q0 - Qs);
#define FM_LOAD2(C, G) do { if constexpr (DUAL) { const unsigned gi_ = (G) < groups ? (G) : groups - 1;                    \
        const float *q_ = q20 - 32 * (int)gi_;                                                                                \
        _Pragma("unroll") for (int u = 0; u < 16; u++) C[u] = q_[2 * (15 - u)]; } } while (0)
#define FM_MMA2(C, B) do { if constexpr (DUAL) { _Pragma("unroll") for (int u = 0; u < 16; u++)                            \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(C[u], B[u], acc2, 0, 0, 0);                                              \
        __builtin_amdgcn_sched_barrier(0); } } while (0)
#define FM_LOAD(A, B, G) do { const unsigned gi_ = (G) < groups ? (G) : groups - 1;   /* clamped: never past the tables */ \
        const float *q_ = q0 - 32 * (int)gi_, *x_ = x0 + (16 * 32 + 16) * (int)gi_;                                          \
        const long long rg_ = rb0 + 32 * (long long)gi_;                                                                      \
        _Pragma("unroll") for (int u = 0; u < 16; u++) {                                                                      \
            A[u] = q_[2 * (15 - u)];                                                                                          \
            if constexpr (GLB) B[u] = gin[((unsigned long long)(rg_ + 2 * u) & row_mask) * M];                                \
            else B[u] = x_[32 * u]; }                                                                                         \
        __builtin_amdgcn_sched_barrier(0); } while (0)   /* keep the loads ahead of the MFMA block that hides them */
#define FM_MMA(A, B) do { _Pragma("unroll") for (int u = 0; u < 16; u++)                                                  \
        acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(B[u], A[u], acc, 0, 0, 0)                                          \
                   : __builtin_amdgcn_mfma_f32_32x32x2f32(A[u], B[u], acc, 0, 0, 0);                                         \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    FM_LOAD2(c0, 0u);
    FM_LOAD(a0, b0, 0u);
    unsigned g = 0;
    for (; g + 2 <= groups; g += 2) {
        FM_LOAD2(c1, g + 1);
        FM_LOAD(a1, b1, g + 1);
        FM_MMA(a0, b0); FM_MMA2(c0, b0);
        FM_LOAD2(c0, g + 2);
        FM_LOAD(a0, b0, g + 2);
        FM_MMA(a1, b1); FM_MMA2(c1, b1);
    }
    if (groups & 1) { FM_MMA(a0, b0); FM_MMA2(c0, b0); }                            // set 0 holds group groups-1 here
#undef FM_LOAD
#undef FM_MMA
#undef FM_LOAD2
#undef FM_MMA2
    if constexpr (DUAL) {
        // second product in the UNTRANSPOSED layout (lane = column (slot, block), register 4g + q = frame 8g + 4kk + q): a
        // register's 32 lanes cover 16 adjacent channels of two rows of the time-major ring
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long t = (GATHER ? s_t0[ch] : T0) + Tj + 8 * (r >> 2) + 4 * kk + (r & 3);
            const unsigned chs = GATHER ? s_ch[ch] : (unsigned)ch;
            if (t < (long)ns) out2_tm[((unsigned long long)(row0 + t) & row_mask) * M + chs] = acc2[r];
        }
    }

    if constexpr (SWAP) {
        // Operands exchanged => D' = D^T (same products, same k order): lane = frame (column lane & 31), register 4g + q =
        // row 8g + 4kk + q = (channel, time block).  A half-wave then stores 32 CONSECUTIVE frames of one channel row: 64
        // contiguous bytes of PCM (128 of float audio) per store instead of 8-byte pieces scattered over 32 rows.
        // (Used when there is no time-major output, whose rows want the untransposed layout.)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i2 = 8 * (r >> 2) + 4 * kk + (r & 3);
            const int sl2 = i2 & 15, blk2 = i2 >> 4;
            const unsigned ch2 = GATHER ? s_ch[sl2] : (unsigned)sl2;
            const long t = (GATHER ? s_t0[sl2] : T0) + 64 * wave + 32 * blk2 + (lane & 31);
            if (t < (long)ns) {
                if (pcm) pcm[(size_t)ch2 * stride + t] = pcm16(acc[r]);
                if (audio) audio[(size_t)ch2 * stride + t] = acc[r];
            }
        }
    } else {
    // D layout: lane holds column j; register g*4+q is row 8g + 4*kk + q  ->  4 consecutive frames per register group
    const bool vec_ok = ((stride & 3) == 0);
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
        const long t = (GATHER ? s_t0[ch] : T0) + Tj + 8 * g4 + 4 * kk;      // frame of register 4g (relative to row0)
        const unsigned chs = GATHER ? s_ch[ch] : (unsigned)ch;
        const float y0 = acc[4 * g4], y1 = acc[4 * g4 + 1], y2 = acc[4 * g4 + 2], y3 = acc[4 * g4 + 3];
        if (t + 3 < (long)ns && vec_ok) {
            if (pcm && ((reinterpret_cast<uintptr_t>(pcm) & 7) == 0)) {
                uint2 w;
                w.x = (unsigned)(uint16_t)pcm16(y0) | ((unsigned)(uint16_t)pcm16(y1) << 16);
                w.y = (unsigned)(uint16_t)pcm16(y2) | ((unsigned)(uint16_t)pcm16(y3) << 16);
                *reinterpret_cast<uint2 *>(pcm + (size_t)chs * stride + t) = w;
            } else if (pcm) {
                int16_t *o = pcm + (size_t)chs * stride + t;
                o[0] = pcm16(y0); o[1] = pcm16(y1); o[2] = pcm16(y2); o[3] = pcm16(y3);
            }
            if (audio && ((reinterpret_cast<uintptr_t>(audio) & 15) == 0))
                *reinterpret_cast<float4 *>(audio + (size_t)chs * stride + t) = make_float4(y0, y1, y2, y3);
            else if (audio) {
                float *o = audio + (size_t)chs * stride + t;
                o[0] = y0; o[1] = y1; o[2] = y2; o[3] = y3;
            }
        } else {
            const float yy[4] = {y0, y1, y2, y3};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (t + q < (long)ns) {
                    if (pcm) pcm[(size_t)chs * stride + t + q] = pcm16(yy[q]);
                    if (audio) audio[(size_t)chs * stride + t + q] = yy[q];
                }
            }
        }
        if (out_tm) {
            const float yy[4] = {y0, y1, y2, y3};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (t + q < (long)ns) out_tm[((unsigned long long)(row0 + t + q) & row_mask) * M + chs] = yy[q];
        }
    }
    }
    }   // active
    if (more) {                                                      // uniform
        __syncthreads();                                             // every wave is done with this tile's window
        commit();
        __syncthreads();
    }
    }   // tiles
}

extern "C" int pmr_fir_mfma_supported(unsigned M, unsigned ntaps)
{
    return M >= 16 && M % 16 == 0 && M <= 16 * 65535u && ntaps >= 2 && (FM_TILE + ntaps + 31) * 4 <= FM_PRE * FM_NT;
}

extern "C" int pmr_launch_fir_mfma(const pmr_switches *sw, pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0,
                                   unsigned ns, unsigned M, const float *taps_pad, unsigned ntaps, float *out_tm, int16_t *pcm,
                                   float *audio, unsigned stride, const unsigned *chan_list, unsigned n_chan,
                                   const float *taps2_pad, float *out2_tm)
{
    if (!ns) return 0;
    if (!pmr_fir_mfma_supported(M, ntaps)) return (int)hipErrorInvalidValue;
    const unsigned qlen = ntaps + 2 * PMR_TAP_PAD, nrows = FM_TILE + ntaps + 31;
    const int dual = taps2_pad && out2_tm && !out_tm;          /* second tap set -> time-major out2_tm, same pass */
    if ((taps2_pad || out2_tm) && !dual) return (int)hipErrorInvalidValue;
    const size_t lds = (dual ? 2 : 1) * (((size_t)qlen + 31) & ~(size_t)31) * sizeof(float) +
                       ((size_t)nrows * 16 + 16 * ((nrows >> 5) + 1)) * sizeof(float);
    /* <= 49 KB of dynamic LDS (window of at most 704 rows x 16 columns + taps): inside the 64 KB default limit */
    hipStream_t st = (hipStream_t)s;
    const unsigned long long rm = (unsigned long long)row_mask;
    const long long r0 = (long long)row0;
    const unsigned tiles = (ns + FM_TILE - 1) / FM_TILE;
    if (chan_list) {
        /* open-channel mask: 16 (channel, segment) units per workgroup */
        if (!n_chan) return 0;
        if (nrows * 16 > 4 * FM_PRE * FM_NT) return (int)hipErrorInvalidValue;
        const unsigned n_units = n_chan * tiles;
        const dim3 grid((n_units + 15) / 16);
        if (dual)
            PMR_KLAUNCH((k_fir_mfma16<false, 1, true, true, true>), grid, dim3(FM_NT), lds, st, in, rm, r0, ns, taps_pad, ntaps,
                               out_tm, pcm, audio, stride, M, chan_list, n_units, tiles, taps2_pad, out2_tm);
        else if (!out_tm)
            PMR_KLAUNCH((k_fir_mfma16<false, 1, true, true>), grid, dim3(FM_NT), lds, st, in, rm, r0, ns, taps_pad, ntaps,
                               out_tm, pcm, audio, stride, M, chan_list, n_units, tiles, (const float *)nullptr, (float *)nullptr);
        else
            PMR_KLAUNCH((k_fir_mfma16<false, 1, false, true>), grid, dim3(FM_NT), lds, st, in, rm, r0, ns, taps_pad, ntaps,
                               out_tm, pcm, audio, stride, M, chan_list, n_units, tiles, (const float *)nullptr, (float *)nullptr);
        return (int)hipGetLastError();
    }
    if (dual) {
        if (nrows * 4 > FM_PRE * FM_NT) return (int)hipErrorInvalidValue;
        PMR_KLAUNCH((k_fir_mfma16<false, 1, true, false, true>), dim3(tiles, M / 16), dim3(FM_NT), lds, st, in, rm, r0, ns,
                           taps_pad, ntaps, out_tm, pcm, audio, stride, M, (const unsigned *)nullptr, 0u, 0u, taps2_pad, out2_tm);
        return (int)hipGetLastError();
    }
    /* PMR_FIR_MFMA=global: B operand straight from the ring (no LDS window; co-resides with front-end tiles).  Measured
     * on MI355X: slower in isolation (0.082 vs 0.066 ms at cfg2) and equal within noise inside the pipelined chain, so the
     * LDS-window kernel stays the default. */
    if (sw->fir_mfma_global) {
        const size_t lds_g = (((size_t)qlen + 31) & ~(size_t)31) * sizeof(float);
        PMR_KLAUNCH((k_fir_mfma16<true, 1, false>), dim3(tiles, M / 16), dim3(FM_NT), lds_g, st, in, rm, r0, ns, taps_pad,
                           ntaps, out_tm, pcm, audio, stride, M, (const unsigned *)nullptr, 0u, 0u, (const float *)nullptr, (float *)nullptr);
        return (int)hipGetLastError();
    }
    /* two tiles per workgroup only while that still leaves enough workgroups to fill the chip (3 per CU fit) */
    const bool two = sw->fir_tpw == 2 && nrows * 4 <= FM_PRE * FM_NT && (size_t)((tiles + 1) / 2) * (M / 16) >= 384;
    if (!two && nrows * 4 > FM_PRE * FM_NT) return (int)hipErrorInvalidValue;
    const dim3 grid(two ? (tiles + 1) / 2 : tiles, M / 16);
#define FM_GO(TPW_, SWAP_) PMR_KLAUNCH((k_fir_mfma16<false, TPW_, SWAP_>), grid, dim3(FM_NT), lds, st, in, rm, r0, ns, taps_pad, \
                                              ntaps, out_tm, pcm, audio, stride, M, (const unsigned *)nullptr, 0u, 0u, (const float *)nullptr, (float *)nullptr)
    if (two) { if (!out_tm) FM_GO(2, true); else FM_GO(2, false); }
    else     { if (!out_tm) FM_GO(1, true); else FM_GO(1, false); }
#undef FM_GO
    return (int)hipGetLastError();
}
