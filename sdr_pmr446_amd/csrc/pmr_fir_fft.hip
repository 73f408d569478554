// pmr_fir_fft.hip -- the audio FIR of large blocks by overlap-save FFT convolution.
//
// reference: firfilt_rrrf_execute_block(ctcss_filt ...) src/sdr_pmr446.c:882 (377-tap high-pass), gain :890, de-emphasis :895-899,
//            PCM hand-off :903-906 / src/dsd_in.c:172-175; the complementary CTCSS low-pass branch delay188(x) - hp(x) :884-889.
//
// Why: demodulating ALL M channels makes this filter the FLOP hot spot of the back end -- 383 MACs per audio sample (gain and the
// truncated de-emphasis response are folded into the taps, pmr_chain.c).  The direct form on the matrix pipe (pmr_fir_mfma4.hip:
// exact k-ordered f32 chains) is bound by the f32 MFMA rate: 0.030 ms per cfg2 block at the 155 TFLOP/s peak,
// 0.055 measured -- a third of the back end's CU time, and the two streams of the chain time-slice the CUs.  The same LINEAR
// filter through a 4096-point FFT costs ~50 vector instructions per output sample instead of 766 FLOP:
//     y = IFFT( FFT(x block) . H ),   H = FFT(taps),   block = N samples, the last N - (ntaps-1) outputs of each are valid,
// i.e. the identical filter, evaluated with different f32 roundings (|error| ~ 6e-7 of the signal scale against 1e-5 allowed;
// int16 PCM within +-1 LSB of the CPU reference as before -- tests/test_gpu_fir_fft.py; the direct MFMA form stays the path of
// small blocks, of the open-channel gather and of the follow-on FIR passes, and `PMR_FIR=direct` selects it everywhere).
//
// Formulation (one workgroup = one PAIR of channels x one block of N frames):
//  * the taps are real, so TWO channels share one complex transform: z = x_a + j x_b, y = h * z = (h * x_a) + j (h * x_b) -- no
//    untangling pass, H is the plain N-point spectrum of the taps;
//  * N = 16 . 16 . R2 (R2 = 4: N = 1024, 64 threads; R2 = 8: N = 2048, 128 threads, round 5; R2 = 16: N = 4096, 256 threads -- the host
//    picks 2048 points where that needs >= 15 % fewer transform points than 1024 and the pass is not DUAL with many channels,
//    pmr_chain.c fir_fft_pick; 4096 points only by PMR_FIR=fft4096), 16 points per thread, three register
//    phases with two LDS exchanges per direction:
//        forward  (decimation in frequency):  A: radix-16 over n0 (stride N/16), x W_N^(r k0)    B: radix-16 over n1, x W_(N/16)^(n2 k1)
//                                             C: radix-R2 over n2 -> X[k0 + 16 k1 + 256 k2] at position k0 N/16 + k1 R2 + k2
//        inverse  (decimation in time, the transposed graph, conjugate twiddles): C', B', A' -> natural order.
//    The forward pass leaves the spectrum digit-reversed, the inverse pass accepts it that way: H is stored in position order by
//    the host (pmr_fir_fft_tables), and phase C, the product with H and phase C' happen in registers without an exchange;
//  * twiddles: exact tables computed in double by the host.  A thread keeps its 15 phase-A factors W_N^(t k0) in registers (they
//    serve A and, conjugated, A'); the 16 x R2 phase-B factors sit in LDS;
//  * LDS: position p at p + (p >> 4) (one float2 of padding per 16): every access pattern of the three phases is conflict-free;
//    34.8 KB per workgroup at N = 4096, 17.4 KB at N = 2048, 8.7 KB at N = 1024 (+ the phase-B twiddles);
//  * loads: lane t of a wave reads the two channels' samples of row (block start + n0 N/16 + t) -- 8 bytes per row; the 8 / 128
//    pair-workgroups that share a row's cache lines are made neighbours on ONE XCD (pmr_xcd_contiguous) so the rows come from HBM
//    once; rows beyond the call's last frame read as zero (whatever the ring holds there never enters the transform);
//  * stores: lane t holds frames (block start + n0 N/16 + t): consecutive lanes = consecutive int16 / float of a channel row;
//  * DUAL (CTCSS detector on): the spectrum is parked in a second LDS buffer and multiplied by a second H (delta_188 - hp) ->
//    second inverse transform -> time-major low-pass ring, same forward transform;
//  * open-channel list (reference :876-877): the pairs are taken from the list (an odd last channel is paired with itself).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

// complex = clang ext-vector pair: element-wise +, -, * and fma map onto v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, the swaps and
// sign flips of a complex product or of a multiplication by +-j onto their op_sel / neg modifiers.  The kernel is VALU-issue-bound
// (tools/ubench/valu_rate.hip: a packed instruction costs 5.3 cycles per wave against 4.7 for a scalar one and does twice the work)
typedef float cf2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ cf2 mk2(float x, float y) { return cf2{x, y}; }
static __device__ __forceinline__ cf2 jtimes(cf2 a) { return cf2{-a.y, a.x}; }                       // j a (constants / rare uses)
// The swaps and sign flips of complex arithmetic as operand modifiers of ONE packed instruction (the compiler materialises them as
// v_mov / v_xor pairs: a third of this kernel's instructions before these helpers).  op_sel / op_sel_hi pick the 32-bit half of
// each 64-bit operand that feeds the low / high result lane, neg_lo / neg_hi negate it.
static __device__ __forceinline__ cf2 add_mj(cf2 a, cf2 b)                                           // a - j b = (a.x + b.y, a.y - b.x)
{
    cf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ cf2 add_pj(cf2 a, cf2 b)                                           // a + j b = (a.x - b.y, a.y + b.x)
{
    cf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ cf2 cmul(cf2 a, cf2 w)                                             // a w   (w: run-time twiddle)
{
    cf2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                   // (a.x w.x, a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));    // + (-a.y w.y, a.y w.x)
    return r;
}
static __device__ __forceinline__ cf2 cmulc(cf2 a, cf2 w)                                            // a conj(w)
{
    cf2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                      // (a.x w.x, -a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));                   // + (a.y w.y, a.y w.x)
    return r;
}
// a w for a COMPILE-TIME w: plain vector code, the swizzled constant folds
static __device__ __forceinline__ cf2 cmulk(cf2 a, cf2 w)
{
    return __builtin_elementwise_fma(cf2{a.y, a.y}, cf2{-w.y, w.x}, cf2{a.x, a.x} * w);
}

// PCM hand-off of a channel PAIR: trunc(x * 32767) toward zero, saturated, NaN -> 0 (src/dsd_in.c:174 + the build's saturation), the
// way the hardware does it by itself: v_cvt_i32_f32 truncates, saturates to the int32 range and turns NaN into 0; v_cvt_pk_i16_i32
// saturates both to int16 and packs them -- 4 instructions per pair where the float clamp (mul, NaN test + select, max, min, cvt) took
// 12 (a sixth of the kernel's vector instructions).  Same integers: clamping before or after the truncation commutes.
typedef short ff_s2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ ff_s2 ff_pcm16x2(cf2 y)
{
    const cf2 s = y * mk2(32767.0f, 32767.0f);
    int ia, ib;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(ia) : "v"(s.x));
    asm("v_cvt_i32_f32 %0, %1" : "=v"(ib) : "v"(s.y));
    return __builtin_bit_cast(ff_s2, __builtin_amdgcn_cvt_pk_i16(ia, ib));
}

// 4-point DFT in place: (a, b, c, d) = x[0..3] -> X[0..3]; forward kernel e^(-j 2 pi nk / 4), INV: e^(+j ...)
template <bool INV>
static __device__ __forceinline__ void r4(cf2 &a, cf2 &b, cf2 &c, cf2 &d)
{
    const cf2 s0 = a + c, s1 = a - c, s2 = b + d, s3 = b - d;
    a = s0 + s2;
    c = s0 - s2;
    b = INV ? add_pj(s1, s3) : add_mj(s1, s3);                     // forward: s1 - j s3
    d = INV ? add_mj(s1, s3) : add_pj(s1, s3);
}

// 16-point DFT of v[0..15] (natural order in).  OUT ORDER: X[k] is left in v[R16P(k)], R16P(k) = 4 (k & 3) + (k >> 2).
#define R16P(k) (4 * ((k) & 3) + ((k) >> 2))
template <bool INV>
static __device__ __forceinline__ void r16(cf2 (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, RH = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; b++) r4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);            // over a (n = 4 a + b): v[4 c + b] = u_b[c]
    // u_b[c] *= W16^(b c)   (conjugated for the inverse)
    const float sg = INV ? 1.f : -1.f;
    const cf2 w1 = mk2(C1, sg * S1), w2 = mk2(RH, sg * RH), w3 = mk2(S1, sg * C1);
    const cf2 w6 = mk2(-RH, sg * RH), w9 = mk2(-C1, -sg * S1);
    v[4 * 1 + 1] = cmulk(v[4 * 1 + 1], w1);                                                  // (c, b) = (1, 1)
    v[4 * 1 + 2] = cmulk(v[4 * 1 + 2], w2);
    v[4 * 1 + 3] = cmulk(v[4 * 1 + 3], w3);
    v[4 * 2 + 1] = cmulk(v[4 * 2 + 1], w2);
    v[4 * 2 + 2] = INV ? jtimes(v[4 * 2 + 2]) : -jtimes(v[4 * 2 + 2]);                      // W16^4 = -j
    v[4 * 2 + 3] = cmulk(v[4 * 2 + 3], w6);
    v[4 * 3 + 1] = cmulk(v[4 * 3 + 1], w3);
    v[4 * 3 + 2] = cmulk(v[4 * 3 + 2], w6);
    v[4 * 3 + 3] = cmulk(v[4 * 3 + 3], w9);
#pragma unroll
    for (int c = 0; c < 4; c++) r4<INV>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);   // over b: v[4 c + d] = X[c + 4 d]
}

// 8-point DFT in place, natural order in AND out: evens / odds by two 4-point DFTs, odd half times W8^k (conjugated for the inverse)
template <bool INV>
static __device__ __forceinline__ void r8(cf2 (&v)[8])
{
    constexpr float RH = 0.70710678118654752f;
    const float sg = INV ? 1.f : -1.f;
    cf2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    r4<INV>(e0, e1, e2, e3);
    r4<INV>(o0, o1, o2, o3);
    o1 = cmulk(o1, mk2(RH, sg * RH));
    o2 = INV ? jtimes(o2) : -jtimes(o2);                             // W8^2 = -j (forward)
    o3 = cmulk(o3, mk2(-RH, sg * RH));
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}

struct ff_params {
    const float *in; unsigned long long row_mask; long long row0; unsigned ns, M;
    const cf2 *H, *H2;                 // [N] spectra of the taps / N, POSITION order (pmr_fir_fft_tables)
    const cf2 *TA;                     // [15][N/16]  W_N^(t k0), k0 = 1..15
    const cf2 *TB;                     // [16][R2]    W_(N/16)^(n2 k1) at [k1][n2]
    int16_t *pcm; float *audio; unsigned stride;
    float *out2_tm;                    // DUAL: time-major ring of the second product
    const unsigned *chan_list; unsigned n_chan, npairs, ntaps;
};

#define FF_IDX(p) ((p) + ((p) >> 4))

// Phases C, x H, C' of one thread, in registers.  PARK: `src` (= zs) holds phase B's output -- run the forward radix-R2 and (DUAL)
// park the spectrum in zs2; !PARK: `src` (= zs2) holds the parked spectrum.  Leaves C' output in zs (own positions only).
// R2 = 16: thread t owns positions 16 t .. 16 t + 15 (k0 = t >> 4, k1 = t & 15); R2 = 4: four runs of 4 (combo c = t + 64 j).
template <int R2, bool DUAL, bool PARK>
static __device__ __forceinline__ void ff_phase_c(cf2 *zs, cf2 *zs2, const cf2 *src, const cf2 *__restrict__ Hp, unsigned t)
{
    if constexpr (R2 == 16) {
        cf2 u[16], w[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) u[n2] = src[FF_IDX(16 * t + n2)];
        if constexpr (PARK) {
            r16<false>(u);
            if constexpr (DUAL) {
#pragma unroll
                for (int k2 = 0; k2 < 16; k2++) zs2[FF_IDX(16 * t + k2)] = u[R16P(k2)];
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) w[k2] = cmul(PARK ? u[R16P(k2)] : u[k2], Hp[16 * t + k2]);
        r16<true>(w);
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) zs[FF_IDX(16 * t + n2)] = w[R16P(n2)];
    } else if constexpr (R2 == 8) {                                // N = 2048: two runs of 8 per thread (combo c = t + 128 j)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const unsigned c = t + 128u * j;
            cf2 u[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; n2++) u[n2] = src[FF_IDX(8 * c + n2)];
            if constexpr (PARK) {
                r8<false>(u);
                if constexpr (DUAL) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) zs2[FF_IDX(8 * c + k2)] = u[k2];
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) u[k2] = cmul(u[k2], Hp[8 * c + k2]);
            r8<true>(u);
#pragma unroll
            for (int n2 = 0; n2 < 8; n2++) zs[FF_IDX(8 * c + n2)] = u[n2];
        }
    } else {
        static_assert(R2 == 4, "radix of the last phase");
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned c = t + 64u * j;
            cf2 u[4];
#pragma unroll
            for (int n2 = 0; n2 < 4; n2++) u[n2] = src[FF_IDX(4 * c + n2)];
            if constexpr (PARK) {
                r4<false>(u[0], u[1], u[2], u[3]);
                if constexpr (DUAL) {
#pragma unroll
                    for (int k2 = 0; k2 < 4; k2++) zs2[FF_IDX(4 * c + k2)] = u[k2];
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) u[k2] = cmul(u[k2], Hp[4 * c + k2]);
            r4<true>(u[0], u[1], u[2], u[3]);
#pragma unroll
            for (int n2 = 0; n2 < 4; n2++) zs[FF_IDX(4 * c + n2)] = u[n2];
        }
    }
}

template <int R2, bool DUAL>
__global__ __launch_bounds__(16 * R2) void k_fir_fft(const ff_params P)
{
    constexpr int N0 = 16 * R2, N = 256 * R2, NT = N0;
    extern __shared__ __attribute__((aligned(16))) char smem_ff[];
    cf2 *zs = reinterpret_cast<cf2 *>(smem_ff);                    // [N + N / 16]
    cf2 *tb = zs + (N + N / 16);                                   // [16][R2]: tb[k1 R2 + n2] (lanes = n2: consecutive addresses)
    cf2 *zs2 = tb + R2 * 16;                                       // DUAL: the parked spectrum, [N + N / 16]
    const unsigned t = threadIdx.x;
    const unsigned unit = pmr_xcd_contiguous(blockIdx.x, gridDim.x);
    const unsigned blk = unit / P.npairs, pair = unit % P.npairs;
    const unsigned L = (unsigned)N - (P.ntaps - 1);                // valid outputs per block
    unsigned ca, cb; bool has_b;
    {
        const unsigned i0 = 2 * pair, i1 = 2 * pair + 1;
        has_b = i1 < P.n_chan;
        ca = P.chan_list ? P.chan_list[i0] : i0;
        cb = has_b ? (P.chan_list ? P.chan_list[i1] : i1) : ca;
    }
    const bool adjacent = cb == ca + 1 && !(ca & 1u);
    // 32-bit index arithmetic: frames relative to row0 (the launcher guarantees ring rows x M < 2^32 elements)
    const int rel0 = (int)(blk * L) - (int)(P.ntaps - 1);          // frame of transform index 0, relative to row0
    const int ns_i = (int)P.ns;
    const unsigned r0lo = (unsigned)P.row0, mask32 = (unsigned)P.row_mask;

    // phase-B twiddles -> LDS, the block's samples -> registers: all in flight together
    for (unsigned i = t; i < (unsigned)(R2 * 16); i += NT) tb[i] = P.TB[i];
    // (the 15 + 15 twiddles of a thread are re-read where they are used -- global table through L1 for A / A', LDS for B / B' --
    //  instead of living in 60 registers across the whole kernel: the register count decides whether a wave of this kernel still
    //  fits on a SIMD beside four front-end tiles)
    // Rows of the transform: INTERIOR blocks (every row lies inside the call and the block does not wrap around the ring: all but the
    // call's last block or two) read row n0 N0 + t at a UNIFORM base (scalar registers) + one per-lane byte offset -- sixteen load
    // instructions and nothing else; the edge blocks test every row (rows beyond the call's last frame read as zero).  Round 4 ran the
    // edge form everywhere: ~15 vector / scalar instructions and two branches per row, a fifth of the kernel.
    const unsigned idx0 = (r0lo + (unsigned)rel0) & mask32;        // ring row of transform index 0 (wave-uniform)
    const bool interior = rel0 + N <= ns_i && (unsigned long long)idx0 + (unsigned)N <= (unsigned long long)mask32 + 1ull;
    cf2 v[16];
    if (interior) {
        const char *rowa = reinterpret_cast<const char *>(P.in) + ((size_t)idx0 * P.M + ca) * sizeof(float);
        const size_t pitch = (size_t)N0 * P.M * sizeof(float);     // bytes between rows n0 and n0 + 1 of a lane
        const unsigned voff = t * P.M * (unsigned)sizeof(float);
        if (adjacent) {
#pragma unroll
            for (int n0 = 0; n0 < 16; n0++) v[n0] = *reinterpret_cast<const cf2 *>(rowa + n0 * pitch + voff);
        } else {
            const long long dba = ((long long)cb - (long long)ca) * (long long)sizeof(float);
#pragma unroll
            for (int n0 = 0; n0 < 16; n0++)
                v[n0] = mk2(*reinterpret_cast<const float *>(rowa + n0 * pitch + voff), *reinterpret_cast<const float *>(rowa + dba + n0 * pitch + voff));
        }
    } else {
#pragma unroll
        for (int n0 = 0; n0 < 16; n0++) {
            const int rel = rel0 + n0 * N0 + (int)t;
            v[n0] = mk2(0.f, 0.f);
            if (rel < ns_i) {
                const float *src = P.in + (size_t)(((r0lo + (unsigned)rel) & mask32) * P.M);
                if (adjacent) v[n0] = *reinterpret_cast<const cf2 *>(src + ca);
                else v[n0] = mk2(src[ca], src[cb]);
            }
        }
    }

    // ---- A: radix-16 over n0, x W_N^(t k0), -> position k0 N0 + t ----
    {
        cf2 wA[16];
#pragma unroll
        for (int k0 = 1; k0 < 16; k0++) wA[k0] = P.TA[(k0 - 1) * N0 + t];
        r16<false>(v);
#pragma unroll
        for (int k0 = 0; k0 < 16; k0++) {
            const cf2 x = k0 ? cmul(v[R16P(k0)], wA[k0]) : v[R16P(0)];
            zs[FF_IDX(k0 * N0 + t)] = x;
        }
    }
    __syncthreads();

    // ---- B: thread (k0, n2): radix-16 over n1 (positions k0 N0 + n1 R2 + n2), x W_N0^(n2 k1), in place ----
    const unsigned k0B = t / R2, n2B = t % R2, baseB = k0B * N0 + n2B;
    {
        cf2 u[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) u[n1] = zs[FF_IDX(baseB + n1 * R2)];
        r16<false>(u);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) zs[FF_IDX(baseB + k1 * R2)] = k1 ? cmul(u[R16P(k1)], tb[k1 * R2 + n2B]) : u[R16P(0)];
    }
    __syncthreads();

    // ---- C, x H, C': in registers (ff_phase_c) ----
    // ---- B', A': the inverse of B and A (conjugate twiddles BEFORE the butterflies), result in v[R16P(n0)] ----
    auto inverse_BA = [&]() {
        __syncthreads();
        {
            cf2 u[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; k1++) {
                const cf2 x = zs[FF_IDX(baseB + k1 * R2)];
                u[k1] = k1 ? cmulc(x, tb[k1 * R2 + n2B]) : x;
            }
            r16<true>(u);
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) zs[FF_IDX(baseB + n1 * R2)] = u[R16P(n1)];
        }
        cf2 wA[16];
#pragma unroll
        for (int k0 = 1; k0 < 16; k0++) wA[k0] = P.TA[(k0 - 1) * N0 + t];      // (in flight across the barrier)
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < 16; k0++) {
            const cf2 x = zs[FF_IDX(k0 * N0 + t)];
            v[k0] = k0 ? cmulc(x, wA[k0]) : x;
        }
        r16<true>(v);
    };

    ff_phase_c<R2, DUAL, true>(zs, zs2, zs, P.H, t);
    inverse_BA();
    // ---- outputs: transform index n = n0 N0 + t is frame rel0 + n (relative to row0); indices >= ntaps - 1 are valid ----
    // Row n0 of a lane is stored by ALL lanes, by none, or (one row per block) by the lanes behind the filter's start-up: decided on
    // the scalar unit per row -- interior blocks carry no per-lane tests except in that one row.
    const int nvalid0 = (int)P.ntaps - 1;                          // first valid transform index
    const auto row_mode = [&](int n0) -> int {                     // 0 skip, 1 every lane, 2 per-lane test
        if (n0 * N0 + (N0 - 1) < nvalid0) return 0;
        if (!interior) return 2;
        return n0 * N0 >= nvalid0 ? 1 : 2;
    };
    {
        // one base pointer per row and thread; the 16 stores of a row are base + n0 N0 (compile-time offsets)
        const long relt = ((long)rel0 + (long)t) & PMR_EXP_PCM_AND;
        int16_t *pa = P.pcm ? P.pcm + (size_t)ca * P.stride + relt : nullptr, *pb = P.pcm ? P.pcm + (size_t)cb * P.stride + relt : nullptr;
        float *aa = P.audio ? P.audio + (size_t)ca * P.stride + relt : nullptr, *ab = P.audio ? P.audio + (size_t)cb * P.stride + relt : nullptr;
#pragma unroll
        for (int n0 = 0; n0 < 16; n0++) {
            const int mode = row_mode(n0);
            if (mode == 0) continue;
            const int n = n0 * N0 + (int)t, rel = rel0 + n;
            if (mode == 1 || (n >= nvalid0 && rel < ns_i)) {
                const cf2 y = v[R16P(n0)];
                if (pa) { const ff_s2 q16 = ff_pcm16x2(y); pa[n0 * N0] = q16.x; if (has_b) pb[n0 * N0] = q16.y; }
                if (aa) { aa[n0 * N0] = y.x; if (has_b) ab[n0 * N0] = y.y; }
            }
        }
    }
    if constexpr (DUAL) {
        __syncthreads();                                           // every thread has read its A' inputs: zs is free again
        ff_phase_c<R2, DUAL, false>(zs, zs2, zs2, P.H2, t);
        inverse_BA();
#pragma unroll
        for (int n0 = 0; n0 < 16; n0++) {
            const int mode = row_mode(n0);
            if (mode == 0) continue;
            const int n = n0 * N0 + (int)t, rel = rel0 + n;
            if (mode == 1 || (n >= nvalid0 && rel < ns_i)) {
                const cf2 y = v[R16P(n0)];
                float *dst = P.out2_tm + (size_t)(((r0lo + (unsigned)rel) & mask32) * P.M);
                if (adjacent) *reinterpret_cast<cf2 *>(dst + ca) = y;
                else { dst[ca] = y.x; if (has_b) dst[cb] = y.y; }
            }
        }
    }
}

// ---- host side: the tables of one transform size, in double ----
// H[p] (position order): p = k0 N0 + k1 R2 + k2  <->  bin k = k0 + 16 k1 + 256 k2;  H[p] = (1/N) sum_d h[d] e^(-j 2 pi k d / N)
extern "C" unsigned pmr_fir_fft_size(int which) { return which == 1 ? 4096u : which == 2 ? 2048u : 1024u; }

extern "C" void pmr_fir_fft_spectrum(unsigned N, const float *h, unsigned ntaps, float *H_out /*[2 N]*/)
{
    const unsigned R2 = N / 256, N0 = N / 16;
    const double w0 = -2.0 * M_PI / (double)N;
    for (unsigned p = 0; p < N; p++) {
        const unsigned k0 = p / N0, k1 = (p % N0) / R2, k2 = p % R2, k = k0 + 16 * k1 + 256 * k2;
        double re = 0.0, im = 0.0;
        for (unsigned d = 0; d < ntaps; d++) {
            const double a = w0 * (double)(((unsigned long long)k * d) % N);
            re += (double)h[d] * cos(a); im += (double)h[d] * sin(a);
        }
        H_out[2 * p] = (float)(re / (double)N); H_out[2 * p + 1] = (float)(im / (double)N);
    }
}

extern "C" void pmr_fir_fft_twiddles(unsigned N, float *TA /*[15][N/16][2]*/, float *TB /*[16][N/256][2]*/)
{
    const unsigned R2 = N / 256, N0 = N / 16;
    for (unsigned k0 = 1; k0 < 16; k0++)
        for (unsigned t = 0; t < N0; t++) {
            const double a = -2.0 * M_PI * (double)((t * k0) % N) / (double)N;
            TA[2 * ((k0 - 1) * N0 + t)] = (float)cos(a); TA[2 * ((k0 - 1) * N0 + t) + 1] = (float)sin(a);
        }
    for (unsigned n2 = 0; n2 < R2; n2++)
        for (unsigned k1 = 0; k1 < 16; k1++) {
            const double a = -2.0 * M_PI * (double)((n2 * k1) % N0) / (double)N0;
            TB[2 * (k1 * R2 + n2)] = (float)cos(a); TB[2 * (k1 * R2 + n2) + 1] = (float)sin(a);
        }
}

extern "C" int pmr_fir_fft_supported(unsigned M, unsigned ntaps)
{
    return M >= 1 && ntaps >= 2 && ntaps <= 512;                   // >= half of the smallest transform stays output
}

// which: 0 = N 1024, 1 = N 4096, 2 = N 2048.  tab: the device tables of that size.
extern "C" int pmr_launch_fir_fft(pmr_stream_t s, int which, const pmr_fir_fft_tab *tab, const float *in, uint64_t row_mask, int64_t row0,
                                  unsigned ns, unsigned M, unsigned ntaps, int16_t *pcm, float *audio, unsigned stride,
                                  float *out2_tm, const unsigned *chan_list, unsigned n_chan)
{
    if (!ns) return 0;
    const unsigned nc = chan_list ? n_chan : M;
    if (!nc) return 0;
    if (!tab || !tab->H || !tab->TA || !tab->TB || (out2_tm && !tab->H2)) return (int)hipErrorInvalidValue;
    const unsigned N = pmr_fir_fft_size(which), L = N - (ntaps - 1);
    ff_params P;
    P.in = in; P.row_mask = (unsigned long long)row_mask; P.row0 = (long long)row0; P.ns = ns; P.M = M;
    P.H = (const cf2 *)tab->H; P.H2 = (const cf2 *)tab->H2; P.TA = (const cf2 *)tab->TA; P.TB = (const cf2 *)tab->TB;
    P.pcm = pcm; P.audio = audio; P.stride = stride; P.out2_tm = out2_tm;
    P.chan_list = chan_list; P.n_chan = nc; P.npairs = (nc + 1) / 2; P.ntaps = ntaps;
    if (((unsigned long long)row_mask + 1ull) * M > 0xffffffffull || (unsigned long long)ns + N > 0x7fffffffull)
        return (int)hipErrorInvalidValue;                                /* the kernel indexes the ring with 32-bit arithmetic */
    const unsigned nblk = (ns + L - 1) / L;
    const unsigned long long units = (unsigned long long)nblk * P.npairs;
    if (units > 0x7fffffffull) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)units);
    const size_t buf = (size_t)(N + N / 16) * sizeof(cf2), tbb = (size_t)(N / 256) * 16 * sizeof(cf2);
    hipStream_t st = (hipStream_t)s;
    if (which == 1) {
        if (out2_tm) {
            static pmr_attr_flags once;
            if (pmr_attr_needed(once))
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_fft<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            PMR_KLAUNCH((k_fir_fft<16, true>), grid, dim3(256), 2 * buf + tbb, st, P);
        } else PMR_KLAUNCH((k_fir_fft<16, false>), grid, dim3(256), buf + tbb, st, P);
    } else if (which == 2) {
        if (out2_tm) PMR_KLAUNCH((k_fir_fft<8, true>), grid, dim3(128), 2 * buf + tbb, st, P);
        else PMR_KLAUNCH((k_fir_fft<8, false>), grid, dim3(128), buf + tbb, st, P);
    } else {
        if (out2_tm) PMR_KLAUNCH((k_fir_fft<4, true>), grid, dim3(64), 2 * buf + tbb, st, P);
        else PMR_KLAUNCH((k_fir_fft<4, false>), grid, dim3(64), buf + tbb, st, P);
    }
    return (int)hipGetLastError();
}
