// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/tile_store tools/exp/tile_store.hip ; run on the GPU box: ./tools/exp/tile_store
// experiment: what does a tile's OUTPUT STORE cost a streaming tile kernel?  The front end's shape without its arithmetic: 32 KB tile by
// LDS-DMA (33.6 KB of LDS: four tiles per CU), a dependent ALU chain standing in for the dc scan + cascade, then 2 KB of output per tile:
//   mode 0  no output
//   mode 1  one 16-byte vector store by half the threads, distinct 2 KB per tile (the front end's level-1 output)
//   mode 2  the same stores, every tile into the same 4 KB window (no HBM write traffic)
//   mode 3  the same stores issued BEFORE the ALU chain (the acknowledgement has the chain's time to arrive)
//   mode 4  scalar stores (s_store_dwordx4 from SGPRs + s_dcache_wb): another path to L2 than the vector memory queue
//   mode 5  mode 1 + an explicit s_waitcnt vmcnt(0) before the end (if waves already wait for their stores this changes nothing)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 4) void k_tile(const char *__restrict__ x, char *__restrict__ out, int chain, int MODE /*run-time: every mode runs the same code*/, unsigned wmask)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int c = blockIdx.x;
    { const int per = gridDim.x >> 3, main = per << 3; if (c < main) c = (c & 7) * per + (c >> 3); }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char *src = x + (size_t)c * 32768 + (size_t)wave * 8192;
    char *dst = smem + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; i++)
        __builtin_amdgcn_global_load_lds((gptr_t *)(src + i * 1024 + lane * 16), (lptr_t *)(dst + i * 1024), 16, 0, 2);
    __syncthreads();
    float4 v = reinterpret_cast<float4 *>(smem)[tid];
    char *o = out + (MODE == 2 ? (size_t)(c & wmask) * 2048 : (size_t)c * 2048);       // mode 2: a window of (wmask + 1) x 2 KB
    if (MODE == 3 && tid < 128) reinterpret_cast<float4 *>(o)[tid] = v;
    // dependent chain: ~chain x 8 cycles per wave (a v_fma per step), with a barrier every 64 steps like the cascade's stages
    float a = v.x;
    for (int k = 0; k < chain; k++) {
        a = fmaf(a, 1.0000001f, v.y);
        if ((k & 63) == 63) __syncthreads();
    }
    v.x = a;
    if (MODE == 1 || MODE == 2 || MODE == 5) { if (tid < 128) reinterpret_cast<float4 *>(o)[tid] = v; }
    if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 4) {
        // 512 bytes per wave: 32 x s_store_dwordx4 of (uniform) SGPR data
        const unsigned s0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a));
        const u4 d = {s0, s0 + 1, s0 + 2, s0 + 3};
        const char *ob = o + wave * 512;
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(d), "s"(ob), "n"(i * 16) : "memory");
        asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE == 0 && a == 1.2345e-30f) reinterpret_cast<float *>(out)[0] = a;
}

template <typename F> static double timeit(F f, int reps = 20)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const size_t bytes = 512ull << 20, ntiles = bytes / 32768;
    char *x, *out;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&out, ntiles * 2048); (void)hipMemset(x, 1, bytes);
    const char *names[6] = {"no output", "16-byte vector store, 2 KB per tile", "the same into one 4 KB window", "vector store BEFORE the ALU chain",
                            "scalar stores + s_dcache_wb", "vector store + s_waitcnt vmcnt(0) at the end"};
    for (int chain : {0, 64, 128, 256}) {
        printf("-- dependent ALU chain of %d steps per wave\n", chain);
#define RUN(M) { double ms = timeit([&] { hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(256), 33600, 0, x, out, chain, M, 1u); }); \
                 printf("   mode %d %-46s %7.4f ms  %6.0f GB/s read\n", M, names[M], ms, bytes / ms / 1e6); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    }
    printf("-- mode 2 by window size (no ALU chain): where does the write stream start to cost?\n");
    for (unsigned wm : {1u, 127u, 1023u, 2047u, 4095u, 8191u, 16383u}) {
        double ms = timeit([&] { hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(256), 33600, 0, x, out, 0, 2, wm); });
        printf("   window %8.2f MB   %7.4f ms\n", (wm + 1) * 2048 / 1048576.0, ms);
    }
    return 0;
}
