// pmr_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the per-block IQ DSP chain.
//
// Stage order and arithmetic follow the reference loop body src/sdr_pmr446.c:795-906 (the liquid-dsp
// objects created at :420-480); the explicit index sums each kernel evaluates are written down and
// checked against the CPU oracle in tests/chain_model.py.  float32 throughout, like liquid's crcf/rrrf.
//
// This file is the STAGED path: one kernel per stage, intermediates in HBM.  It is the correctness
// baseline that the fused front-end (pmr_frontend.hip) is A/B-tested against.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pmr_kernels.h"

typedef float2 cf;

static __device__ __forceinline__ cf cf_make(float r, float i) { cf v; v.x = r; v.y = i; return v; }

// ------------------------------------------------------------------------------------------------
// DC blocker  H(z) = (1 - z^-1) / (1 + a1 z^-1),  a1 = -1 + alpha          (:422, :795)
//   direct form II as liquid runs it:  v0 = x - a1*v1 ;  y = v0 - v1
// The recurrence is a first-order linear scan: inside a 4096-sample tile every thread runs 16 samples
// serially, then a decayed Hillis-Steele scan (multipliers lambda^(16*2^j)) gives each thread its carry.
// Across tiles the same scan runs on tile aggregates (pmr_launch_dc_scan).
// ------------------------------------------------------------------------------------------------

template <bool APPLY>
__global__ __launch_bounds__(256) void k_dcblock(const cf *__restrict__ x, unsigned n_in,
                                                 const cf *__restrict__ W, cf *__restrict__ agg_out,
                                                 cf *__restrict__ out, pmr_dc_consts c,
                                                 const float *__restrict__ lam_thread_pow)
{
    __shared__ cf sh[256];
    const unsigned tile = blockIdx.x, t = threadIdx.x;
    const size_t base = (size_t)tile * PMR_DC_TILE + (size_t)t * 16;
    cf xs[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        size_t n = base + j;
        xs[j] = n < n_in ? x[n] : cf_make(0.f, 0.f);
    }
    // local recurrence from zero state
    float vr = 0.f, vi = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        vr = __fsub_rn(xs[j].x, __fmul_rn(c.a1, vr));
        vi = __fsub_rn(xs[j].y, __fmul_rn(c.a1, vi));
    }
    sh[t] = cf_make(vr, vi);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned d = 1u << j;
        cf tmp = t >= d ? sh[t - d] : cf_make(0.f, 0.f);
        __syncthreads();
        if (t >= d) {
            cf cur = sh[t];
            cur.x = fmaf(c.lam_pow16[j], tmp.x, cur.x);
            cur.y = fmaf(c.lam_pow16[j], tmp.y, cur.y);
            sh[t] = cur;
        }
        __syncthreads();
    }
    if (!APPLY) {
        if (t == 255) agg_out[tile] = sh[255];
        return;
    }
    // v just before this thread's first sample
    cf w = W[tile];
    cf ex = t > 0 ? sh[t - 1] : cf_make(0.f, 0.f);
    const float lp = lam_thread_pow[t];
    float v1r = fmaf(lp, w.x, ex.x), v1i = fmaf(lp, w.y, ex.y);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        float v0r = __fsub_rn(xs[j].x, __fmul_rn(c.a1, v1r));
        float v0i = __fsub_rn(xs[j].y, __fmul_rn(c.a1, v1i));
        size_t n = base + j;
        if (n < n_in) out[n] = cf_make(__fsub_rn(v0r, v1r), __fsub_rn(v0i, v1i));
        v1r = v0r; v1i = v0i;
    }
}

__global__ __launch_bounds__(1024) void k_dc_scan(const cf *__restrict__ agg, unsigned ntiles,
                                                  cf *__restrict__ W, cf *__restrict__ state,
                                                  pmr_dc_consts c, const float *__restrict__ idx_pow,
                                                  float lam_last, float inv_last)
{
    __shared__ cf sh[1024];
    __shared__ cf carry_s;
    const unsigned t = threadIdx.x;
    if (t == 0) carry_s = *state;
    __syncthreads();
    for (unsigned base = 0; base < ntiles; base += 1024) {
        const unsigned i = base + t;
        const cf a = i < ntiles ? agg[i] : cf_make(0.f, 0.f);
        const cf carry = carry_s;
        sh[t] = a;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const unsigned d = 1u << j;
            cf tmp = t >= d ? sh[t - d] : cf_make(0.f, 0.f);
            __syncthreads();
            if (t >= d) {
                cf cur = sh[t];
                cur.x = fmaf(c.lam_tile_pow[j], tmp.x, cur.x);
                cur.y = fmaf(c.lam_tile_pow[j], tmp.y, cur.y);
                sh[t] = cur;
            }
            __syncthreads();
        }
        const cf inc = sh[t];
        const cf ex = t > 0 ? sh[t - 1] : cf_make(0.f, 0.f);
        const float ip = idx_pow[t];
        const cf Wi = cf_make(fmaf(ip, carry.x, ex.x), fmaf(ip, carry.y, ex.y));
        if (i < ntiles) W[i] = Wi;
        if (i == ntiles - 1) {
            // v after the last VALID sample: undo the decay over the zero padding of the last tile
            *state = cf_make(fmaf(lam_last, Wi.x, inv_last * a.x), fmaf(lam_last, Wi.y, inv_last * a.y));
        }
        __syncthreads();
        if (t == 1023) carry_s = inc;     // (lambda^4096)^1024 underflows to 0: older carry is gone
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Half-band decimator stage: z1[i] = z0[2i+1-2m] + sum_j h1[j] * z0[2i - 2(2m-1-j)]   
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halfband(const cf *__restrict__ zin, cf *__restrict__ zout,
                                                  unsigned n_out, int keep_in, int par, int m,
                                                  const float *__restrict__ h1, float scale)
{
    const unsigned o = blockIdx.x * 256u + threadIdx.x;
    if (o >= n_out) return;
    const cf *p = zin + keep_in + 2 * (long)o - par;   // -> absolute index 2i
    float yr = 0.f, yi = 0.f;
    const int L = 2 * m;
    for (int j = 0; j < L; j++) {
        const cf s = p[-2 * (L - 1 - j)];
        const float h = h1[j];
        yr = fmaf(h, s.x, yr);
        yi = fmaf(h, s.y, yi);
    }
    const cf d = p[1 - 2 * m];
    zout[o] = cf_make((d.x + yr) * scale, (d.y + yi) * scale);
}

// ------------------------------------------------------------------------------------------------
// Arbitrary polyphase resampler with 24-bit phase: out[j] = sum_k bank[idx_j][k] * dec[q_j - 13 + k]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_arb(const cf *__restrict__ dec, cf *__restrict__ out,
                                             unsigned long long out_pos0, unsigned long long out_mask, unsigned ny,
                                             uint32_t phase0, uint32_t step, const float *__restrict__ bank,
                                             int keep)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= ny) return;
    const uint64_t ph = (uint64_t)phase0 + (uint64_t)j * step;
    const long q = (long)(ph >> 24);
    const unsigned idx = (unsigned)(ph & 0xffffffu) >> 16;
    const float *b = bank + idx * 14u;
    const cf *p = dec + keep + q - 13;
    float yr = 0.f, yi = 0.f;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const cf s = p[k];
        yr = fmaf(b[k], s.x, yr);
        yi = fmaf(b[k], s.y, yi);
    }
    out[(out_pos0 + j) & out_mask] = cf_make(yr, yi);
}

// ------------------------------------------------------------------------------------------------
// NCO shift (:808-812) + polyphase analysis filter bank + M-point FFT (:814) + discriminator (:881).
// One workgroup owns FT frames in LDS (the first one is the frame BEFORE its range, recomputed so the
// discriminator has conj(prev) without a cross-workgroup dependency).
// ------------------------------------------------------------------------------------------------
// frames held in LDS per workgroup (one of them is the recomputed previous frame): at most 64 KB worth, and few enough
// that a block still yields ~1000 workgroups (the work per frame is small; parallelism is what matters)
static __host__ __device__ inline unsigned chan_ft_auto(unsigned M, unsigned ns)
{
    // LDS: X[FT][M] complex -- 32 KB per tile, 64 KB for M >= 512 (taller tiles amortise the 25-frame filter history)
    unsigned cap = (M >= 512u ? 8192u : 4096u) / M; if (cap < 2u) cap = 2u;
    // many channels: about one tile per CU at least (measured at cfg5, 838 frames x 1024 channels: FT 4..6 is 25 % faster
    // than FT 2); few channels: the cap
    unsigned want = M >= 256u ? ns / 256u + 1u : cap;
    if (want < 2u) want = 2u;
    return want < cap ? want : cap;
}

template <int F, int P>
static __device__ __forceinline__ void pfb_rows(const pmr_chan_params &q, unsigned log2M, cf *Xs, long long fbase,
                                                unsigned nfl /*frames to store*/, unsigned tid)
{
    const unsigned M = q.M, nco_mask = q.nco_period - 1;
    const cf *__restrict__ xr = (const cf *)q.xr;
    const cf *__restrict__ nco_cs = (const cf *)q.nco_cs;
    // work item = (channel c, group g of F consecutive frames); M >= 256: one group, a thread walks channels;
    // small M: the 256 threads are 256/M groups side by side (16 lanes = one 128-byte row at M = 16)
    const unsigned G = (nfl + F - 1) / F;
    typedef float v2 __attribute__((ext_vector_type(2)));       // (re, im): real-tap MACs map onto v_pk_fma_f32
    const unsigned xr_mask32 = (unsigned)q.xr_mask;
    // low 32 bits of the absolute sample index are all the ring / NCO masks need (indices before the stream start wrap
    // into the zero-initialised top of the ring)
    const unsigned abase = (unsigned)((unsigned long long)fbase * (unsigned long long)M);
    for (unsigned w = tid; w < M * G; w += 256) {
        const unsigned c = w & (M - 1), f0 = (w >> log2M) * F;
        float h[P];
#pragma unroll
        for (int k = 0; k < P; k++) h[k] = q.taps_t[k * M + c];
        v2 acc[F];
#pragma unroll
        for (int f = 0; f < F; f++) acc[f] = v2{0.f, 0.f};
        const unsigned a0 = abase + f0 * M + c;
        constexpr int RB = 8;                                    // rows per batch: loads first, then the MACs
#pragma unroll
        for (int r0 = 0; r0 < F + P - 1; r0 += RB) {
            v2 xm[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int r = r0 + u;
                if (r < F + P - 1) {
                    const unsigned a = a0 + (unsigned)r * M;
                    const cf x = xr[a & xr_mask32];
                    const cf cs = nco_cs[a & nco_mask];
                    xm[u] = v2{fmaf(x.x, cs.x, x.y * cs.y), fmaf(x.y, cs.x, -(x.x * cs.y))};   // x * conj(e^{j theta})
                }
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int r = r0 + u;
                if (r < F + P - 1) {
#pragma unroll
                    for (int f = (r - P + 1 > 0 ? r - P + 1 : 0); f <= (r < F - 1 ? r : F - 1); f++)
                        acc[f] = __builtin_elementwise_fma(v2{h[r - f], h[r - f]}, xm[u], acc[f]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned rc = __brev(c) >> (32 - log2M);
#pragma unroll
        for (int f = 0; f < F; f++)
            if (f0 + f < nfl) Xs[(f0 + f) * M + rc] = cf_make(acc[f].x, acc[f].y);
    }
}

__global__ __launch_bounds__(256) void k_channelize(pmr_chan_params q, unsigned log2M, unsigned FT)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned M = q.M, p = q.p, ns = q.ns;
    const unsigned TFN = FT - 1;
    cf *Xs = reinterpret_cast<cf *>(smem);            // [FT][M]
    cf *tw = Xs + (size_t)FT * M;                     // [M/2]
    const cf *__restrict__ xr = (const cf *)q.xr;
    const cf *__restrict__ nco_cs = (const cf *)q.nco_cs;
    const unsigned nco_mask = q.nco_period - 1;
    const unsigned tid = threadIdx.x;
    const unsigned t0 = blockIdx.x * TFN;             // first NEW frame of this tile (relative to frame0)
    const unsigned nf = min(TFN, ns - t0);            // new frames in this tile

    for (unsigned k = tid; k < M / 2; k += 256) tw[k] = ((const cf *)q.fft_tw)[k];

    // phase 1: X[f][c] = sum_k taps_t[k][c] * xm[(F + f - p + k) * M + c],  local f = 0 is frame t0-1
    const long long fbase = (long long)q.frame0 + t0 - (long long)p;   // absolute frame of (f = 0, k = 0)
    if (p == 26) {
        // sliding window: a thread owns one channel and F consecutive frames, so a resampled sample is loaded (and
        // NCO-mixed) once per group instead of once per tap -- (F + 25) loads for F frames instead of 3 * 26 * F
        if (FT > 8)      pfb_rows<16, 26>(q, log2M, Xs, fbase, nf + 1, tid);
        else if (FT > 4) pfb_rows<8, 26>(q, log2M, Xs, fbase, nf + 1, tid);
        else             pfb_rows<4, 26>(q, log2M, Xs, fbase, nf + 1, tid);
    } else {
    const unsigned items = (nf + 1) * M;
    for (unsigned w = tid; w < items; w += 256) {
        const unsigned f = w >> log2M, c = w & (M - 1);
        float ar = 0.f, ai = 0.f;
        for (unsigned k = 0; k < p; k++) {
            const long long a = (fbase + f + k) * (long long)M + c;    // absolute resampled sample index
            const cf x = xr[(unsigned long long)a & q.xr_mask];
            const cf cs = nco_cs[(unsigned)a & nco_mask];
            const float xmr = fmaf(x.x, cs.x, x.y * cs.y);       // x * conj(e^{j theta})
            const float xmi = fmaf(x.y, cs.x, -(x.x * cs.y));
            const float h = q.taps_t[k * M + c];
            ar = fmaf(h, xmr, ar);
            ai = fmaf(h, xmi, ai);
        }
        const unsigned rc = __brev(c) >> (32 - log2M);
        Xs[f * M + rc] = cf_make(ar, ai);
    }
    }
    __syncthreads();

    // phase 2: in-place radix-2 DIT FFT of every frame (forward, unscaled)
    const unsigned nb = (nf + 1) * (M / 2);
    for (unsigned len = 2, lg = 1; len <= M; len <<= 1, lg++) {
        const unsigned half = len >> 1, tstep = M >> lg;
        for (unsigned b = tid; b < nb; b += 256) {
            const unsigned f = b >> (log2M - 1), r = b & (M / 2 - 1);
            const unsigned grp = r >> (lg - 1), k = r & (half - 1);
            const unsigned i0 = f * M + grp * len + k, i1 = i0 + half;
            const cf wv = tw[k * tstep];
            const cf a = Xs[i0], bb = Xs[i1];
            const float tr = fmaf(bb.x, wv.x, -(bb.y * wv.y));
            const float ti = fmaf(bb.x, wv.y, bb.y * wv.x);
            Xs[i0] = cf_make(a.x + tr, a.y + ti);
            Xs[i1] = cf_make(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }

    // phase 3: discriminator m = arg(conj(prev) * cur) / (2 pi kf), plus tap-offs
    cf *__restrict__ chan_out = (cf *)q.chan_out;
    const unsigned oitems = nf * M;
    for (unsigned w = tid; w < oitems; w += 256) {
        const unsigned f = w >> log2M, k = w & (M - 1);
        const cf pv = Xs[f * M + k], cu = Xs[(f + 1) * M + k];
        const float re = fmaf(pv.x, cu.x, pv.y * cu.y);
        const float im = fmaf(pv.x, cu.y, -(pv.y * cu.x));
        const unsigned long long row = (unsigned long long)(q.frame0 + t0 + f) & q.fm_mask;
        const bool rst = t0 + f == 0 && q.reset_flags && q.reset_flags[k];   // freqdem_reset: previous sample = 0 -> arg(0) = 0
        q.fm[row * M + k] = rst ? 0.f : pmr_arg(im, re) * q.fm_ref;
        if (chan_out) chan_out[(size_t)k * q.chan_stride + t0 + f] = cu;
    }
    if (q.rssi_part) {
        for (unsigned k = tid; k < M; k += 256) {
            float a = 0.f;
            for (unsigned f = 0; f < nf; f++) {
                const cf cu = Xs[(f + 1) * M + k];
                a += hypotf(cu.x, cu.y);
            }
            q.rssi_part[(size_t)blockIdx.x * M + k] = a;
        }
    }
}

__global__ void k_rssi_finish(const float *__restrict__ part, unsigned ntiles, unsigned M, unsigned ns,
                              float *__restrict__ rssi_db)
{
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    float a = 0.f;
    for (unsigned t = 0; t < ntiles; t++) a += part[(size_t)t * M + k];
    rssi_db[k] = 20.f * log10f(a / (float)ns);       // average_power(), :330-336
}

typedef float v2f __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// k_fir_pair: the direct-form audio FIR for channel counts the MFMA kernel does not take (M not a multiple of 16) -- time-major real
// FIR y[t] = sum_d h[d] x[t-d], accumulated oldest sample first, + epilogue (gain :890, de-emphasis IIR :898, int16 PCM
// src/dsd_in.c:174 / float audio :904), PACKED ACROSS TWO ADJACENT CHANNELS.  lane <-> (channel pair, 16-output
// segment); accumulator i is the v2f (channel 2c, channel 2c+1) of output t0 - J + i, so one v_pk_fma_f32 per
// (step, output) does two MACs with the tap broadcast from a single SGPR (no SGPR-pair alignment, no moves) and
// the input pair is one 8-byte load from the time-major stream.  v_pk_fma_f32 sustains ~1.8x the FLOP rate of
// v_fma_f32 on gfx950 (tools/ubench/valu_rate.hip).  taps_c is h zero-padded by PMR_TAP_PAD on both sides.
// ------------------------------------------------------------------------------------------------
#define FP_R 16
#define FP_J 0                                 /* no IIR warm-up: the host folds gain + de-emphasis into the taps */
#define FP_RP (FP_R + FP_J)

static __device__ __forceinline__ int16_t pcm_from_float(float y)
{
    const float s = y * 32767.0f;
    if (!(s == s)) return 0;
    if (s >= 32767.0f) return 32767;
    if (s <= -32768.0f) return -32768;
    return (int16_t)s;                                 // truncation toward zero (src/dsd_in.c:174)
}

__global__ __launch_bounds__(256) void k_fir_pair(const float *__restrict__ in, unsigned long long row_mask,
                                                  long long row0, unsigned ns, unsigned M,
                                                  unsigned log2Mh, const float *__restrict__ taps_c,
                                                  unsigned ntaps, float gain, int iir, float b0, float b1, float a1,
                                                  float *__restrict__ out_tm, int16_t *__restrict__ pcm,
                                                  float *__restrict__ audio, unsigned stride)
{
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned cp = gid & ((M >> 1) - 1), seg = gid >> log2Mh;
    const long t0 = (long)seg * FP_R;
    if (t0 >= (long)ns) return;
    v2f acc[FP_RP];
#pragma unroll
    for (int i = 0; i < FP_RP; i++) acc[i] = v2f{0.f, 0.f};
    // step e brings input frame s = t0 - J - (ntaps-1) + e; it meets accumulator i with tap h[ntaps-1 + i - e]
    const long long r0 = row0 + t0 - (long long)FP_J - (long long)(ntaps - 1);
    const float *pc = in + 2 * cp;
#define FP_ROW(e) (pc + ((unsigned long long)(r0 + (e)) & row_mask) * M)
    const float *tq0 = taps_c + PMR_TAP_PAD + (ntaps - 1);
    const unsigned steps = ntaps + FP_RP - 1;
    unsigned e = 0;
    for (; e + 4 <= steps; e += 4) {
        const v2f x0 = *reinterpret_cast<const v2f *>(FP_ROW(e + 0));
        const v2f x1 = *reinterpret_cast<const v2f *>(FP_ROW(e + 1));
        const v2f x2 = *reinterpret_cast<const v2f *>(FP_ROW(e + 2));
        const v2f x3 = *reinterpret_cast<const v2f *>(FP_ROW(e + 3));
        const float *tp = tq0 - (long)e - 3;           // tap(e + u, i) = tp[3 - u + i]: one window for 4 steps
#pragma unroll
        for (int i = 0; i < FP_RP; i++) {
            v2f a = acc[i];
            a = __builtin_elementwise_fma(v2f{tp[3 + i], tp[3 + i]}, x0, a);
            a = __builtin_elementwise_fma(v2f{tp[2 + i], tp[2 + i]}, x1, a);
            a = __builtin_elementwise_fma(v2f{tp[1 + i], tp[1 + i]}, x2, a);
            a = __builtin_elementwise_fma(v2f{tp[i], tp[i]}, x3, a);
            acc[i] = a;
        }
    }
    for (; e < steps; e++) {
        const v2f x = *reinterpret_cast<const v2f *>(FP_ROW(e));
        const float *tp = tq0 - (long)e;
#pragma unroll
        for (int i = 0; i < FP_RP; i++) acc[i] = __builtin_elementwise_fma(v2f{tp[i], tp[i]}, x, acc[i]);
    }
    // epilogue per channel of the pair: gain (:890) -> de-emphasis IIR (:898) -> float audio (:904) / int16 PCM
    float ya[FP_R], yb[FP_R];
    {
        float va = 0.f, vb = 0.f;
#pragma unroll
        for (int i = 0; i < FP_RP; i++) {
            float ua = __fmul_rn(acc[i].x, gain), ub = __fmul_rn(acc[i].y, gain);
            float oa = ua, ob = ub;
            if (iir) {
                const float wa = __fsub_rn(ua, __fmul_rn(a1, va)), wb = __fsub_rn(ub, __fmul_rn(a1, vb));
                oa = __fadd_rn(__fmul_rn(b0, wa), __fmul_rn(b1, va));
                ob = __fadd_rn(__fmul_rn(b0, wb), __fmul_rn(b1, vb));
                va = wa; vb = wb;
            }
            if (i >= FP_J) { ya[i - FP_J] = oa; yb[i - FP_J] = ob; }
        }
    }
    const unsigned ka = 2 * cp, kb = 2 * cp + 1;
    const bool full = t0 + FP_R <= (long)ns;
    if (out_tm) {
#pragma unroll
        for (int i = 0; i < FP_R; i++)
            if (t0 + i < (long)ns)
                *reinterpret_cast<v2f *>(out_tm + ((unsigned long long)(row0 + t0 + i) & row_mask) * M + ka) = v2f{ya[i], yb[i]};
    }
    if (audio) {
        float *oa = audio + (size_t)ka * stride + t0, *ob = audio + (size_t)kb * stride + t0;
        if (full && ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(audio) & 15) == 0)) {
#pragma unroll
            for (int i = 0; i < FP_R; i += 4) {
                *reinterpret_cast<float4 *>(oa + i) = make_float4(ya[i], ya[i + 1], ya[i + 2], ya[i + 3]);
                *reinterpret_cast<float4 *>(ob + i) = make_float4(yb[i], yb[i + 1], yb[i + 2], yb[i + 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < FP_R; i++) if (t0 + i < (long)ns) { oa[i] = ya[i]; ob[i] = yb[i]; }
        }
    }
    if (pcm) {
        int16_t *oa = pcm + (size_t)ka * stride + t0, *ob = pcm + (size_t)kb * stride + t0;
        if (full && ((stride & 7) == 0) && ((reinterpret_cast<uintptr_t>(pcm) & 15) == 0)) {
            // 16 consecutive int16 per channel = two 16-byte stores
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned wa[4], wb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = 8 * h + 2 * j;
                    wa[j] = (unsigned)(uint16_t)pcm_from_float(ya[i]) | ((unsigned)(uint16_t)pcm_from_float(ya[i + 1]) << 16);
                    wb[j] = (unsigned)(uint16_t)pcm_from_float(yb[i]) | ((unsigned)(uint16_t)pcm_from_float(yb[i + 1]) << 16);
                }
                *reinterpret_cast<uint4 *>(oa + 8 * h) = make_uint4(wa[0], wa[1], wa[2], wa[3]);
                *reinterpret_cast<uint4 *>(ob + 8 * h) = make_uint4(wb[0], wb[1], wb[2], wb[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < FP_R; i++)
                if (t0 + i < (long)ns) { oa[i] = pcm_from_float(ya[i]); ob[i] = pcm_from_float(yb[i]); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ingest formats other than cf32 (include/pmr_io.h): interleaved int16 I/Q scaled by 1/32768, interleaved
// uint8 I/Q (rtl_sdr) as (x - 127.5) / 127.5 -- the same rules as the host-side reader in pmr_io.c.  Converting on the
// device means 4 or 2 bytes per sample cross PCIe instead of 8.  Four samples per thread.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_iq_convert(const void *__restrict__ raw, float4 *__restrict__ out, unsigned n_in, int fmt)
{
    const unsigned i4 = blockIdx.x * 256u + threadIdx.x, i = 4u * i4;            // samples [i, i + 4)
    if (i >= n_in) return;
    float v[8];
    if (i + 4 <= n_in) {
        if (fmt == 1) {
            const uint4 w = reinterpret_cast<const uint4 *>(raw)[i4];
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[2 * k] = (float)(short)(ww[k] & 0xffffu) * (1.0f / 32768.0f);
                v[2 * k + 1] = (float)(short)(ww[k] >> 16) * (1.0f / 32768.0f);
            }
        } else {
            const uint2 w = reinterpret_cast<const uint2 *>(raw)[i4];
            const unsigned ww[2] = {w.x, w.y};
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = ((float)((ww[k >> 2] >> (8 * (k & 3))) & 0xffu) - 127.5f) * (1.0f / 127.5f);
        }
        out[2 * i4] = make_float4(v[0], v[1], v[2], v[3]);
        out[2 * i4 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        float *o = reinterpret_cast<float *>(out);
        for (unsigned j = 2 * i; j < 2 * n_in; j++)
            o[j] = fmt == 1 ? (float)reinterpret_cast<const short *>(raw)[j] * (1.0f / 32768.0f)
                            : ((float)reinterpret_cast<const unsigned char *>(raw)[j] - 127.5f) * (1.0f / 127.5f);
    }
}

extern "C" int pmr_launch_iq_convert(pmr_stream_t s, const void *raw, void *out_cf32, unsigned n_in, int fmt)
{
    if (!n_in) return 0;
    PMR_KLAUNCH(k_iq_convert, dim3((n_in + 1023) / 1024), dim3(256), 0, (hipStream_t)s, raw, (float4 *)out_cf32, n_in, fmt);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline unsigned ilog2(unsigned v) { unsigned l = 0; while ((1u << l) < v) l++; return l; }

extern "C" int pmr_launch_dc_agg(pmr_stream_t s, const void *x, unsigned n_in, void *agg,
                                 const pmr_dc_consts *c, const float *lam_thread_pow)
{
    const unsigned ntiles = (n_in + PMR_DC_TILE - 1) / PMR_DC_TILE;
    if (!ntiles) return 0;
    PMR_KLAUNCH(k_dcblock<false>, dim3(ntiles), dim3(256), 0, (hipStream_t)s, (const cf *)x, n_in,
                       (const cf *)nullptr, (cf *)agg, (cf *)nullptr, *c, lam_thread_pow);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_dc_scan(pmr_stream_t s, const void *agg, unsigned ntiles, void *W, void *state,
                                  const pmr_dc_consts *c, const float *lam_tile_idx_pow, float lam_last,
                                  float inv_last)
{
    if (!ntiles) return 0;
    PMR_KLAUNCH(k_dc_scan, dim3(1), dim3(PMR_DC_SCAN_THREADS), 0, (hipStream_t)s, (const cf *)agg, ntiles,
                       (cf *)W, (cf *)state, *c, lam_tile_idx_pow, lam_last, inv_last);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_dc_apply(pmr_stream_t s, const void *x, unsigned n_in, const void *W, void *out,
                                   const pmr_dc_consts *c, const float *lam_thread_pow)
{
    const unsigned ntiles = (n_in + PMR_DC_TILE - 1) / PMR_DC_TILE;
    if (!ntiles) return 0;
    PMR_KLAUNCH(k_dcblock<true>, dim3(ntiles), dim3(256), 0, (hipStream_t)s, (const cf *)x, n_in,
                       (const cf *)W, (cf *)nullptr, (cf *)out, *c, lam_thread_pow);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_halfband(pmr_stream_t s, const void *zin, void *zout, unsigned n_out, int keep_in,
                                   int par, int m, const float *h1, float scale)
{
    if (!n_out) return 0;
    PMR_KLAUNCH(k_halfband, dim3((n_out + 255) / 256), dim3(256), 0, (hipStream_t)s, (const cf *)zin,
                       (cf *)zout, n_out, keep_in, par, m, h1, scale);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_arb(pmr_stream_t s, const void *dec, void *out_ring, uint64_t out_pos0, uint64_t out_mask,
                              unsigned ny, uint32_t phase0, uint32_t step, const float *bank, int keep)
{
    if (!ny) return 0;
    PMR_KLAUNCH(k_arb, dim3((ny + 255) / 256), dim3(256), 0, (hipStream_t)s, (const cf *)dec, (cf *)out_ring,
                       (unsigned long long)out_pos0, (unsigned long long)out_mask, ny, phase0, step, bank, keep);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_channelize(pmr_stream_t s, const pmr_chan_params *p, unsigned *ntiles_out)
{
    const unsigned ft = chan_ft_auto(p->M, p->ns);                      /* tile height: chosen on the host, handed to the kernel */
    const unsigned ntiles = (p->ns + ft - 2) / (ft - 1);
    if (ntiles_out) *ntiles_out = ntiles;
    if (!p->ns) return 0;
    const size_t lds = ((size_t)ft * p->M + p->M / 2) * sizeof(cf);
    static pmr_attr_flags attr_set{0};
    if (pmr_attr_needed(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_channelize),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    PMR_KLAUNCH(k_channelize, dim3(ntiles), dim3(256), lds, (hipStream_t)s, *p, ilog2(p->M), ft);
    return (int)hipGetLastError();
}

extern "C" int pmr_launch_rssi_finish(pmr_stream_t s, const float *rssi_part, unsigned ntiles, unsigned M,
                                      unsigned ns, float *rssi_db)
{
    PMR_KLAUNCH(k_rssi_finish, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)s, rssi_part, ntiles, M, ns,
                       rssi_db);
    return (int)hipGetLastError();
}

/* second tap set in the same pass (CTCSS low-pass branch): the MFMA kernel only; -1: caller runs two passes */
extern "C" int pmr_launch_fir_dual(pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0, unsigned ns,
                                   unsigned M, const float *taps_pad, const float *taps2_pad, unsigned ntaps, int16_t *pcm, float *audio,
                                   unsigned stride, float *out2_tm, const unsigned *chan_list, unsigned n_chan)
{
    if (!pmr_fir_mfma4_supported(M, ntaps)) return -1;
    return pmr_launch_fir_mfma4(s, in, row_mask, row0, ns, M, taps_pad, ntaps, nullptr, pcm, audio, stride, chan_list, n_chan, taps2_pad, out2_tm,
                                nullptr, nullptr);
}

extern "C" int pmr_launch_fir_tm(pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0,
                                 unsigned ns, unsigned M, const float *taps_pad, unsigned ntaps, float gain, int iir, float b0,
                                 float b1, float a1, float *out_tm, int16_t *pcm, float *audio, unsigned stride,
                                 const unsigned *chan_list, unsigned n_chan, const pmr_rssi_job *job, int *job_done)
{
    if (job_done) *job_done = 0;
    if (!ns) return 0;
    /* the direct form: on the matrix pipe (pmr_fir_mfma4.hip) where M is a multiple of 16, else packed over channel pairs on the VALU */
    if (gain == 1.0f && !iir && pmr_fir_mfma4_supported(M, ntaps))
        return pmr_launch_fir_mfma4(s, in, row_mask, row0, ns, M, taps_pad, ntaps, out_tm, pcm, audio, stride, chan_list, n_chan, nullptr, nullptr,
                                    job, job_done);
    if (chan_list || M < 2) return (int)hipErrorInvalidValue;            /* (the open-channel list needs the MFMA kernel: pmr_chain_set_channel_mask) */
    const unsigned segs = (ns + FP_R - 1) / FP_R;
    const size_t threads = (size_t)segs * (M >> 1);
    PMR_KLAUNCH(k_fir_pair, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s, in,
                       (unsigned long long)row_mask, (long long)row0, ns, M,
                       ilog2(M >> 1), taps_pad, ntaps, gain, iir, b0, b1, a1, out_tm, pcm, audio, stride);
    return (int)hipGetLastError();
}
