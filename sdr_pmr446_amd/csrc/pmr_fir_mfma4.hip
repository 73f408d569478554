// pmr_fir_mfma4.hip -- the DIRECT form of the audio FIR (reference src/sdr_pmr446.c:882-904: 377-tap CTCSS high-pass, gain, 50 us
// de-emphasis, PCM hand-off; the complementary CTCSS low-pass branch :884-889; the optional FIR de-emphasis / low-pass passes
// :896,:901) on the gfx950 matrix pipe with v_mfma_f32_16x16x4_f32, 16 channels x 128 frames per workgroup.
//
// Where it runs (round 4): small blocks (the reference's 100 000-sample calls), the open-channel gather, the follow-on FIR passes,
// and everything under PMR_FIR=direct -- large blocks of the default chain take the overlap-save FFT form (pmr_fir_fft.hip), whose
// results this kernel bounds in tests/test_gpu_fir_fft.py.
//
// Formulation: banded Toeplitz x data, taps with gain and the truncated de-emphasis response folded in, exact k-ordered f32
// accumulation oldest sample first = liquid's firfilt order.
//   * D = A B with the 16x16x4 shape: A[i][kappa] = g[i + (n-1) - kappa] (lane l: row l & 15, kappa = 4 s + (l >> 4)),
//     B[kappa][j] = X[T - (n-1) + kappa][channel j] (lane l: 64 CONSECUTIVE floats of the time-major window per step: no bank
//     conflicts, no padding).  The band needs n - 1 + 16 kappa per 16 frames: 398 -> 100 steps;
//   * a wave owns 32 frames: two accumulators (16 frames each) that share every A operand and alternate on the pipe (32-cycle
//     issue, 40-cycle dependent latency);
//   * a workgroup = 4 waves = 128 frames: 33 KB of LDS, four per CU, 2731 workgroups per cfg2 block;
//   * the epilogue goes through LDS so that PCM / audio leave as whole 16-byte pieces of a channel row (256 B per channel).
// Measured on MI355X (round 3): the k loop alone reaches 140-150 TFLOP/s with 2-4 workgroups per CU (tools/exp/mfma_f32_rate.hip:
// 90-96 % of the f32 MFMA peak, operands from LDS), the kernel 88 (cfg2: 0.053 ms): the rest is the window traffic -- 2731 x 36 KB
// = 98 MB per cfg2 block.  Round 2's 32x32x2 / 256-frame form (pmr_fir_mfma.hip, bit-identical results, the faster one in the
// cfg2 chain of round 3) and forms that slid the window over several tiles were retired in round 4 with the FFT form's arrival.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define F4_NT 256
#define F4_TILE 128                              /* frames per workgroup: 4 waves x 2 accumulators x 16 */
#define F4_GS 8                                  /* k-steps per software-pipeline group */
#define F4_STAGE 10                              /* float4 per thread to stage a whole window: up to 640 rows */
#define F4_LDY (F4_TILE + 4)                     /* epilogue staging: floats per channel row */

static __device__ __forceinline__ int16_t pcm16_4(float y)
{
    // branch-free (four of these sit in every epilogue store): NaN -> 0, saturation by clamping, truncation toward zero by the cast
    float s = y * 32767.0f;
    s = s == s ? s : 0.f;
    s = __builtin_fminf(__builtin_fmaxf(s, -32768.0f), 32767.0f);
    return (int16_t)(int)s;                       // truncation toward zero (src/dsd_in.c:174), saturated
}

// The k loop of one wave: two 16-frame accumulators (frames 32 wave + 16 a + i of the window's tile) over `ngroups` groups of F4_GS
// steps.  step s of accumulator a: A = Q[PAD + (ntaps-1) + (lane & 15) - (lane >> 4) - 4 s], B = Xs[(32 wave + 16 a + 4 s) * 16 + lane].
// Two register sets: group g+1's operands are in flight from LDS while group g's MFMAs issue.
template <bool DUAL>
static __device__ __forceinline__ void f4_kloop(const float *Qs, const float *Q2, const float *Xs, unsigned ntaps, unsigned ngroups,
                                                int wave, int lane, f32x4 &acc0, f32x4 &acc1, f32x4 &acd0, f32x4 &acd1)
{
    const float *qa = Qs + PMR_TAP_PAD + (ntaps - 1) + (lane & 15) - (lane >> 4) - 4 * (F4_GS - 1);   // group 0: step u at qa[4 (GS-1-u)]
    const float *q2 = Q2 + (qa - Qs);
    const float *xb = Xs + (32 * wave) * 16 + lane;                                                    // group 0: step u at xb[64 u] (+256: acc 1)
    float a0[F4_GS], b00[F4_GS], b01[F4_GS], a1[F4_GS], b10[F4_GS], b11[F4_GS];
    float c0[DUAL ? F4_GS : 1], c1[DUAL ? F4_GS : 1];
#define F4_LOAD(A, B0, B1, C, G) do { const unsigned gi_ = (G) < ngroups ? (G) : ngroups - 1;    /* clamped: never past the tables */ \
    const float *q_ = qa - 4 * F4_GS * (int)gi_, *x_ = xb + 64 * F4_GS * (int)gi_;                                                 \
    _Pragma("unroll") for (int u = 0; u < F4_GS; u++) {                                                                             \
        A[u] = q_[4 * (F4_GS - 1 - u)]; B0[u] = x_[64 * u]; B1[u] = x_[256 + 64 * u];                                               \
        if constexpr (DUAL) C[u] = (q2 - 4 * F4_GS * (int)gi_)[4 * (F4_GS - 1 - u)]; }                                              \
    __builtin_amdgcn_sched_barrier(0); } while (0)   /* keep the loads ahead of the MFMA block that hides them */
#define F4_MMA(A, B0, B1, C) do { _Pragma("unroll") for (int u = 0; u < F4_GS; u++) {                                                  \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u], B0[u], acc0, 0, 0, 0);                                                    \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u], B1[u], acc1, 0, 0, 0);                                                    \
        if constexpr (DUAL) {                                                                                                       \
            acd0 = __builtin_amdgcn_mfma_f32_16x16x4f32(C[u], B0[u], acd0, 0, 0, 0);                                                \
            acd1 = __builtin_amdgcn_mfma_f32_16x16x4f32(C[u], B1[u], acd1, 0, 0, 0); } }                                            \
    __builtin_amdgcn_sched_barrier(0); } while (0)
    F4_LOAD(a0, b00, b01, c0, 0u);
    unsigned g = 0;
    for (; g + 2 <= ngroups; g += 2) {
        F4_LOAD(a1, b10, b11, c1, g + 1);
        F4_MMA(a0, b00, b01, c0);
        F4_LOAD(a0, b00, b01, c0, g + 2);
        F4_MMA(a1, b10, b11, c1);
    }
    if (ngroups & 1) F4_MMA(a0, b00, b01, c0);                       // set 0 holds group ngroups-1 here
#undef F4_LOAD
#undef F4_MMA
}

// time-major store of one wave's two accumulators: register q of lane l = frame t0 + 16 a + 4 (l >> 4) + q, channel (l & 15)
static __device__ __forceinline__ void f4_store_tm(float *__restrict__ out, unsigned long long row_mask, long long row0, unsigned M,
                                                   unsigned ns, long t0, unsigned chs, int lane, const f32x4 &d0, const f32x4 &d1)
{
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const f32x4 d = a ? d1 : d0;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const long t = t0 + 16 * a + 4 * (lane >> 4) + qq;
            if (t < (long)ns) out[((unsigned long long)(row0 + t) & row_mask) * M + chs] = d[qq];
        }
    }
}

// GATHER (open-channel mask, reference :876-877): the 16 columns of a tile are 16 arbitrary (channel, 128-frame segment) units.
// DUAL: a second tap set over the same samples -> time-major out2_tm (the CTCSS low-pass branch, :884-889), same pass.
// TMOUT: the product itself also goes out time-major (intermediate of the FIR de-emphasis / low-pass chains).
template <bool GATHER, bool DUAL, bool TMOUT>
__global__ __launch_bounds__(F4_NT, 4) void k_fir_mfma4(const float *__restrict__ in, unsigned long long row_mask, long long row0,
                                                      unsigned ns, const float *__restrict__ taps_c, unsigned ntaps,
                                                      float *__restrict__ out_tm, int16_t *__restrict__ pcm,
                                                      float *__restrict__ audio, unsigned stride, unsigned M,
                                                      const unsigned *__restrict__ chan_list, unsigned n_units, unsigned nseg,
                                                      const float *__restrict__ taps2_c, float *__restrict__ out2_tm, pmr_rssi_job job)
{
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if constexpr (!GATHER) {
        const unsigned L = pmr_xcd_contiguous(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        bx = L % gridDim.x; by = L / gridDim.x;
    }
    if (job.rssi_db) {
        // the rider (launcher: small blocks only): one extra workgroup -- the last of a 1-D grid, workgroup 0 of an extra row of
        // a 2-D one -- sums the channelizer tiles' partial |y| sums in tile order (k_rssi_finish, pmr_kernels.hip)
        const bool rider = GATHER ? blockIdx.x == gridDim.x - 1 : by == gridDim.y - 1;
        if (rider) {
            if (GATHER || bx == 0) {
                for (unsigned k = threadIdx.x; k < job.M; k += F4_NT) {
                    float a = 0.f;
                    for (unsigned t = 0; t < job.ntiles; t++) a += job.rssi_part[(size_t)t * job.M + k];
                    job.rssi_db[k] = 20.f * log10f(a / (float)job.ns);       // average_power(), :330-336
                }
            }
            return;
        }
    }
    __shared__ unsigned s_ch[16];                                    // channel of column slot s
    __shared__ long s_t0[16];                                        // first frame of slot s
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (GATHER) {
        if (tid < 16) {
            const unsigned u = blockIdx.x * 16u + tid;
            s_ch[tid] = chan_list[u < n_units ? u / nseg : 0];
            s_t0[tid] = u < n_units ? (long)(u % nseg) * F4_TILE : (long)ns;          // beyond the block: nothing stored
        }
    } else {
        const unsigned cg0 = by * 16u;                               // a group of 16 adjacent channels per workgroup row
        in += cg0;
        if (out_tm) out_tm += cg0;
        if (DUAL) out2_tm += cg0;
        if (pcm) pcm += (size_t)cg0 * stride;
        if (audio) audio += (size_t)cg0 * stride;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_4[];
    const unsigned qlen = ntaps + 2 * PMR_TAP_PAD, qpad = (qlen + 31) & ~31u;
    float *Qs = reinterpret_cast<float *>(smem_4);                   // padded taps
    float *Q2 = Qs + qpad;                                           // DUAL: second padded tap table
    float *Xs = Q2 + (DUAL ? qpad : 0u);                             // window: row r at 16 r (r <-> frame T0 - (ntaps-1) + r)
    const unsigned nsteps = (ntaps - 1 + 16 + 3) / 4, ngroups = (nsteps + F4_GS - 1) / F4_GS;
    const unsigned nrows = (F4_TILE - 16) + 4 * F4_GS * ngroups;
    const long T0 = GATHER ? 0 : (long)bx * F4_TILE;                 // first frame of the tile (relative to row0)

    for (unsigned i = tid; i < qlen; i += F4_NT) Qs[i] = taps_c[i];
    if constexpr (DUAL) for (unsigned i = tid; i < qlen; i += F4_NT) Q2[i] = taps2_c[i];

    // ---- window: HBM ring -> LDS.  All loads of a thread are issued before the first is stored ----
    if constexpr (GATHER) {
        __syncthreads();                                             // slot tables
        for (unsigned e0 = tid; e0 < nrows * 16; e0 += F4_NT * 11) {
            float v[11];
#pragma unroll
            for (int i = 0; i < 11; i++) {
                const unsigned e = e0 + F4_NT * i, r = e >> 4, sl = e & 15;
                const long t = s_t0[sl] - (long)(ntaps - 1) + r;
                v[i] = 0.f;
                if (e < nrows * 16 && t < (long)ns) v[i] = in[((unsigned long long)(row0 + t) & row_mask) * M + s_ch[sl]];
            }
#pragma unroll
            for (int i = 0; i < 11; i++) {
                const unsigned e = e0 + F4_NT * i;
                if (e < nrows * 16) Xs[e] = v[i];
            }
        }
    } else {
        {
            float4 v[F4_STAGE];                                      // nrows * 4 <= F4_STAGE * F4_NT (pmr_fir_mfma4_supported)
#pragma unroll
            for (int i = 0; i < F4_STAGE; i++) {
                const unsigned u = tid + F4_NT * i, r = u >> 2, q4 = (u & 3) * 4;
                const long t = T0 - (long)(ntaps - 1) + r;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u < nrows * 4 && t < (long)ns)
                    v[i] = *reinterpret_cast<const float4 *>(in + ((unsigned long long)(row0 + t) & row_mask) * M + q4);
            }
#pragma unroll
            for (int i = 0; i < F4_STAGE; i++) {
                const unsigned u = tid + F4_NT * i;
                if (u < nrows * 4) reinterpret_cast<float4 *>(Xs)[u] = v[i];
            }
        }
    }
    __syncthreads();

    // ---- banded Toeplitz x window ----
    const bool active = GATHER || T0 + 32 * wave < (long)ns;         // else: the whole wave lies beyond the block
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acd0 = acc0, acd1 = acc0;
    if (active) f4_kloop<DUAL>(Qs, Q2, Xs, ntaps, ngroups, wave, lane, acc0, acc1, acd0, acd1);

    // D layout: lane holds column (lane & 15) = channel slot; register q = row 4 (lane >> 4) + q = frame
    const int sl = lane & 15, fr = 32 * wave + 4 * (lane >> 4);      // + 16 a + q
    if (active) {
        const long tw = (GATHER ? s_t0[sl] : T0) + 32 * wave;
        const unsigned chs = GATHER ? s_ch[sl] : (unsigned)sl;
        if constexpr (DUAL) f4_store_tm(out2_tm, row_mask, row0, M, ns, tw, chs, lane, acd0, acd1);
        if constexpr (TMOUT) f4_store_tm(out_tm, row_mask, row0, M, ns, tw, chs, lane, acc0, acc1);
    }
    if (!pcm && !audio) return;                                      // uniform

    // ---- channel-major outputs through LDS: Ys[slot][frame], then 8 consecutive frames of one channel per thread ----
    __syncthreads();                                                 // every wave is done with the window
    float *Ys = Xs;                                                  // 16 x F4_LDY floats (8.4 KB) over the window
    if (active) {
        *reinterpret_cast<f32x4 *>(Ys + sl * F4_LDY + fr) = acc0;
        *reinterpret_cast<f32x4 *>(Ys + sl * F4_LDY + fr + 16) = acc1;
    }
    __syncthreads();
    {
        const int os = tid >> 4, f0 = 8 * (tid & 15);                // slot, first frame inside the tile
        const long tb = (GATHER ? s_t0[os] : T0) + f0;
        const unsigned chs = GATHER ? s_ch[os] : (unsigned)os;
        if (tb < (long)ns) {
            const f32x4 y0 = *reinterpret_cast<const f32x4 *>(Ys + os * F4_LDY + f0);
            const f32x4 y1 = *reinterpret_cast<const f32x4 *>(Ys + os * F4_LDY + f0 + 4);
            const float yy[8] = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
            const bool full = tb + 8 <= (long)ns;
            if (pcm) {
                int16_t *o = pcm + (size_t)chs * stride + tb;
                if (full && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
                    uint4 w;
                    w.x = (unsigned)(uint16_t)pcm16_4(yy[0]) | ((unsigned)(uint16_t)pcm16_4(yy[1]) << 16);
                    w.y = (unsigned)(uint16_t)pcm16_4(yy[2]) | ((unsigned)(uint16_t)pcm16_4(yy[3]) << 16);
                    w.z = (unsigned)(uint16_t)pcm16_4(yy[4]) | ((unsigned)(uint16_t)pcm16_4(yy[5]) << 16);
                    w.w = (unsigned)(uint16_t)pcm16_4(yy[6]) | ((unsigned)(uint16_t)pcm16_4(yy[7]) << 16);
                    *reinterpret_cast<uint4 *>(o) = w;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) if (tb + i < (long)ns) o[i] = pcm16_4(yy[i]);
                }
            }
            if (audio) {
                float *o = audio + (size_t)chs * stride + tb;
                if (full && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
                    *reinterpret_cast<f32x4 *>(o) = y0;
                    *reinterpret_cast<f32x4 *>(o + 4) = y1;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) if (tb + i < (long)ns) o[i] = yy[i];
                }
            }
        }
    }
}

extern "C" int pmr_fir_mfma4_supported(unsigned M, unsigned ntaps)
{
    /* the window (112 + 32 ceil((ntaps + 18) / 32) rows of 64 B) + the tap table(s) must fit four workgroups' LDS budget */
    const unsigned nsteps = (ntaps - 1 + 16 + 3) / 4, ngroups = (nsteps + F4_GS - 1) / F4_GS;
    return M >= 16 && M % 16 == 0 && M <= 16 * 65535u && ntaps >= 2 && ((F4_TILE - 16) + 4 * F4_GS * ngroups) * 4 <= F4_STAGE * F4_NT;
}

extern "C" int pmr_launch_fir_mfma4(pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0, unsigned ns, unsigned M,
                                    const float *taps_pad, unsigned ntaps, float *out_tm, int16_t *pcm, float *audio, unsigned stride,
                                    const unsigned *chan_list, unsigned n_chan, const float *taps2_pad, float *out2_tm,
                                    const pmr_rssi_job *job, int *job_done)
{
    if (job_done) *job_done = 0;
    if (!ns) return 0;
    if (!pmr_fir_mfma4_supported(M, ntaps)) return (int)hipErrorInvalidValue;
    const int dual = taps2_pad && out2_tm;
    if ((taps2_pad || out2_tm) && !dual) return (int)hipErrorInvalidValue;
    const unsigned qlen = ntaps + 2 * PMR_TAP_PAD, qpad = (qlen + 31) & ~31u;
    const unsigned nsteps = (ntaps - 1 + 16 + 3) / 4, ngroups = (nsteps + F4_GS - 1) / F4_GS;
    const unsigned nrows = (F4_TILE - 16) + 4 * F4_GS * ngroups;
    size_t win = (size_t)nrows * 16;
    if (win < 16 * F4_LDY) win = 16 * F4_LDY;                       /* the epilogue's staging lives in the window's space */
    const size_t lds = ((dual ? 2 : 1) * (size_t)qpad + win) * sizeof(float);
    hipStream_t st = (hipStream_t)s;
    const unsigned long long rm = (unsigned long long)row_mask;
    const long long r0 = (long long)row0;
    const unsigned tiles = (ns + F4_TILE - 1) / F4_TILE;
    dim3 grid(tiles, M / 16);
    unsigned n_units = 0;
    if (chan_list) {
        if (!n_chan) return 0;
        n_units = n_chan * tiles;
        grid = dim3((n_units + 15) / 16);
    }
    pmr_rssi_job jb = {nullptr, 0, 0, 0, nullptr};
    if (job && job->rssi_db && job->rssi_part && tiles <= 64) {     /* the rider costs a workgroup (1-D grid) or a row of idle ones (2-D) */
        jb = *job;
        if (chan_list) grid.x += 1; else grid.y += 1;
        if (job_done) *job_done = 1;
    }
#define F4_GO(G_, D_, T_) PMR_KLAUNCH((k_fir_mfma4<G_, D_, T_>), grid, dim3(F4_NT), lds, st, in, rm, r0, ns, taps_pad, ntaps, out_tm, \
                                             pcm, audio, stride, M, chan_list, n_units, tiles, taps2_pad, out2_tm, jb)
    if (chan_list) {
        if (dual) { if (out_tm) F4_GO(true, true, true); else F4_GO(true, true, false); }
        else      { if (out_tm) F4_GO(true, false, true); else F4_GO(true, false, false); }
    } else {
        if (dual) { if (out_tm) F4_GO(false, true, true); else F4_GO(false, true, false); }
        else      { if (out_tm) F4_GO(false, false, true); else F4_GO(false, false, false); }
    }
#undef F4_GO
    return (int)hipGetLastError();
}
