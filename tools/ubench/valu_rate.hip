// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (VGPR and SGPR multiplier) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1, int iters)
{
    float x = threadIdx.x * 1e-3f;
    if (MODE == 0) {            // scalar fma, 16 independent accumulators, SGPR multiplier
        float a[16];
        for (int i = 0; i < 16; i++) a[i] = i;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fmaf(s0, x, a[i]);
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fmaf(s1, x, a[i]);
        }
        float r = 0; for (int i = 0; i < 16; i++) r += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    } else {                     // packed fma, 16 independent v2f accumulators
        v2f a[16];
        for (int i = 0; i < 16; i++) a[i] = v2f{(float)i, (float)-i};
        v2f xx = v2f{x, x + 1.f};
        v2f m0 = (MODE == 1) ? v2f{s0, s1} : v2f{x * 0.5f, x * 0.25f};
        v2f m1 = (MODE == 1) ? v2f{s1, s0} : v2f{x * 0.125f, x * 2.f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_elementwise_fma(m0, xx, a[i]);
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_elementwise_fma(m1, xx, a[i]);
        }
        v2f r = v2f{0, 0}; for (int i = 0; i < 16; i++) r += a[i];
        out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
    }
}

template <int MODE> void run(const char *name, int wg_per_cu)
{
    float *out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    const int iters = 4000, grid = 256 * wg_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 256>>>(out, 1.0001f, 0.9999f, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<grid, 256>>>(out, 1.0001f, 0.9999f, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)grid * 4 /*waves*/ * iters * 32;             // wave-instructions
    double per_simd = instr / 1024.0;
    printf("%-28s wg/cu %d: %.3f ms  -> %.2f cycles per wave-instr per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", name, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / per_simd, instr * 64 * (MODE == 0 ? 2 : 4) / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32 (sgpr mult)", w);
        run<1>("v_pk_fma_f32 (sgpr pair)", w);
        run<2>("v_pk_fma_f32 (vgpr)", w);
    }
    return 0;
}
