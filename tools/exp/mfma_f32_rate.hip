// tools/exp/mfma_f32_rate.hip -- what limits the audio FIR's matrix-pipe rate?  The FIR kernels (pmr_fir_mfma*.hip) reach ~57 % of the
// f32 MFMA peak.  This micro-benchmark runs their k-loop shape in isolation: MODE 0 operands in registers (pure issue rate),
// MODE 1 operands re-read from LDS every step (ds_read_b32, the FIR's pattern), with W workgroups of 256 threads per CU.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip && ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE /*4: 16x16x4, 2: 32x32x2*/, int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, const float *src)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int u = 0; u < 8; u++) { a[u] = lds[lane + 64 * u]; b[u] = lds[4096 + lane + 64 * u]; }
    if constexpr (SHAPE == 4) {
        f32x4 acc[NACC];
        for (int n = 0; n < NACC; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; it++) {
            if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < 8; u++) { a[u] = lds[((it * 8 + u) * 4 + (lane >> 4)) & 4095]; b[u] = lds[4096 + ((it * 512 + 64 * u + lane) & 4095)]; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + n) & 7], acc[n], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int n = 0; n < NACC; n++) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x16 acc[NACC];
        for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) acc[n][i] = 0.f;
        for (int it = 0; it < iters; it++) {
            if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < 8; u++) { a[u] = lds[((it * 8 + u) * 2 + (lane >> 5)) & 4095]; b[u] = lds[4096 + ((it * 256 + 32 * u + lane) & 4095)]; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + n) & 7], acc[n], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) s += acc[n][i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int SHAPE, int MODE, int NACC>
static void run(const char *name, int wg_per_cu, float *out, const float *src)
{
    const int ncu = 256, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, MODE, NACC>), dim3(ncu * wg_per_cu), dim3(256), 32768 + (wg_per_cu == 1 ? 65536 : 0), 0, out, iters, src);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)ncu * wg_per_cu * 4 /*waves*/ * iters * 8.0 * NACC * (SHAPE == 4 ? 2048.0 : 4096.0);
        if (rep) printf("%-34s wg/CU %d  %.3f ms  %.1f TF/s\n", name, wg_per_cu, ms, flop / ms / 1e9);
    }
}

int main()
{
    float *out, *src; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&src, 8192 * 4);
    std::vector<float> h(8192); for (int i = 0; i < 8192; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k<4, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 1; w <= 4; w++) {
        if (w == 1) {
            hipFuncSetAttribute((const void *)k<4, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void *)k<4, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void *)k<4, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void *)k<2, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void *)k<2, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void *)k<2, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        run<4, 0, 2>("16x16x4 regs   2 acc", w, out, src);
        run<4, 1, 2>("16x16x4 LDS    2 acc", w, out, src);
        run<4, 0, 4>("16x16x4 regs   4 acc", w, out, src);
        run<4, 1, 4>("16x16x4 LDS    4 acc", w, out, src);
        run<2, 0, 1>("32x32x2 regs   1 acc", w, out, src);
        run<2, 1, 1>("32x32x2 LDS    1 acc", w, out, src);
        run<2, 0, 2>("32x32x2 regs   2 acc", w, out, src);
    }
    return 0;
}
