// pmr_carry_load.hpp -- the front end's dc carry subtracted where a channelizer LOADS the resampled stream (pmr_carry_fix,
// pmr_kernels.h; reference stages src/sdr_pmr446.c:795-796 feeding :804-814).
//
// Output j of the block belongs to front-end tile c = q / TQ, q = (phi0 + j * step) >> 24 (the decimated sample the resampler
// stood at), and misses  V_c * ((Kgain * GA[branch]) * mu^q'),  q' = q - c * TQ + HhQ,  branch = bits 16..23 of the phase: exactly
// k_fe_tilefix's expression, evaluated in the same order, so a sample corrected here equals the sample corrected in place.
//
// A thread walks samples at a constant stride (one per frame row), so everything is INCREMENTAL: the 64-bit phase advances by
// stride * step, the tile index by at most NOV tiles per row (compare-and-subtract, no division), and the tables (GA, mu^q', the
// few carries a workgroup can meet) sit in LDS.  Decimated indices carry a bias of nbias tiles so that history samples in
// front of the block (j < 0, never corrected) keep the arithmetic non-negative.
#ifndef PMR_CARRY_LOAD_HPP
#define PMR_CARRY_LOAD_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmr_kernels.h"

typedef float pmr_cfv __attribute__((ext_vector_type(2)));

struct pmr_carry_lds {              // workgroup-uniform: LDS tables + constants
    const float *GA, *G12;          // [256], [TQ + HhQ (+ slack)]
    const pmr_cfv *V;               // [nv] carries of tiles c_lo .. c_lo + nv - 1 (zero outside the block's tiles)
    int c_lo;                       // biased index of the first tile of the table
};

static __host__ __device__ inline unsigned pmr_carry_lds_floats(const pmr_carry_fix &f)
{
    return 256u + ((f.TQ + f.HhQ + 3u) & ~3u) + 2u * f.nv;
}

// biased tile index of output j (any sign): floor((((phi0 + j * step) >> 24) + qbias) / TQ)
static __device__ __forceinline__ int pmr_carry_tile(const pmr_carry_fix &f, long long j, int *qp_out)
{
    const long long ph = (long long)f.phi0 + j * (long long)f.step;
    const int qp = (int)(ph >> 24) + (int)f.qbias;             // >= 0 by the choice of the bias
    if (qp_out) *qp_out = qp;
    return (int)((unsigned)qp / f.TQ);
}

// workgroup set-up: tables -> LDS.  j_first = block-relative index of the LOWEST sample any thread of the workgroup loads.
// Call from every thread; a __syncthreads() must follow before the tables are used.
template <int NT>
static __device__ __forceinline__ pmr_carry_lds pmr_carry_setup(const pmr_carry_fix &f, float *lds, long long j_first, int tid)
{
    pmr_carry_lds t;
    float *ga = lds, *g12 = lds + 256;
    const unsigned n12 = f.TQ + f.HhQ;
    pmr_cfv *v = reinterpret_cast<pmr_cfv *>(g12 + ((n12 + 3u) & ~3u));
    t.GA = ga; t.G12 = g12; t.V = v;
    t.c_lo = pmr_carry_tile(f, j_first, nullptr);
    for (unsigned i = tid; i < 256u; i += NT) ga[i] = f.GA[i];
    for (unsigned i = tid; i < n12; i += NT) g12[i] = f.G12[i];
    for (unsigned i = tid; i < f.nv; i += NT) {
        const long long c = (long long)t.c_lo + i - (long long)f.nbias;
        v[i] = (c >= 0 && c < (long long)f.ntiles) ? ((const pmr_cfv *)f.V)[c] : pmr_cfv{0.f, 0.f};
    }
    return t;
}

struct pmr_carry_state {            // per thread
    unsigned long long ph;          // phi0 + j * step of the sample the next call corrects (two's complement for j < 0)
    int j;                          // its block-relative index
    int qb;                         // (biased tile) * TQ - HhQ - qbias: q' = low32(ph >> 24) - qb
    int vi;                         // tile - c_lo
};

static __device__ __forceinline__ pmr_carry_state pmr_carry_init(const pmr_carry_fix &f, const pmr_carry_lds &t, long long j)
{
    pmr_carry_state s;
    int qp;
    const int c = pmr_carry_tile(f, j, &qp);
    s.ph = (unsigned long long)((long long)f.phi0 + j * (long long)f.step);
    s.j = (int)j;
    s.qb = c * (int)f.TQ - (int)f.HhQ - (int)f.qbias;
    s.vi = c - t.c_lo;
    return s;
}

// correct sample x (output s.j) and advance the state by one row: dj outputs, dph = dj * step.
// Out-of-range samples (history in front of the block, the tail corrected in place) get a zero gain: x - V * 0 = x.
template <int NOV>
static __device__ __forceinline__ pmr_cfv pmr_carry_apply(const pmr_carry_fix &f, const pmr_carry_lds &t, pmr_carry_state &s,
                                                          pmr_cfv x, int dj, unsigned long long dph)
{
#ifdef PMR_CARRY_NOOP       /* timing experiment only (tools/variant_bench.sh): what would the chain gain if this cost nothing?  WRONG results */
    return x;
#endif
    const unsigned lo = (unsigned)s.ph;
    const int q = (int)__builtin_amdgcn_alignbit((unsigned)(s.ph >> 32), lo, 24);       // low 32 bits of ph >> 24
    int ql = q - s.qb;
    const int TQ = (int)f.TQ, TQH = (int)(f.TQ + f.HhQ);
#pragma unroll
    for (int o = 0; o < NOV; o++) {
        const int d = ql >= TQH ? TQ : 0;
        ql -= d; s.qb += d;
        s.vi += d ? 1 : 0;
    }
    float gg = t.GA[(lo >> 16) & 0xffu] * t.G12[ql];
    gg = (unsigned)s.j < f.fix_limit ? gg : 0.f;
    const pmr_cfv V = t.V[s.vi];
    x = pmr_cfv{fmaf(-V.x, gg, x.x), fmaf(-V.y, gg, x.y)};
    s.ph += dph;
    s.j += dj;
    return x;
}

#endif
