// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/cumask_probe tools/exp/cumask_probe.hip ; run on the GPU box: ./tools/exp/cumask_probe
// experiment: which (XCC, SE, CU) does bit i of a stream's CU mask (hipExtStreamCreateWithCUMask) select on this part?
// A stream is created with the low `nbits` bits set; a kernel of many short workgroups records XCC_ID / HW_ID of where it ran.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void k_where(unsigned *out)
{
    // HW_REG_XCC_ID = 20 (bits 3:0), HW_REG_HW_ID = 4 (gfx9: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    // stay a little so that the workgroups spread over every CU the mask allows
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) { }
}

static void probe(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", name, hipGetErrorString(e)); return; }
    const int nb = 8192;
    unsigned *d; (void)hipMalloc(&d, 2 * nb * sizeof(unsigned));
    (void)hipMemsetAsync(d, 0xff, 2 * nb * sizeof(unsigned), st);
    hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, st, d);
    std::vector<unsigned> h(2 * nb);
    (void)hipMemcpyAsync(h.data(), d, 2 * nb * sizeof(unsigned), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    std::set<unsigned> xccs; std::set<unsigned> cus;
    unsigned per_xcc[16] = {0};
    std::set<unsigned> cu_of_xcc[16];
    for (int b = 0; b < nb; b++) {
        const unsigned xcc = h[2 * b] & 15, hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        xccs.insert(xcc); cu_of_xcc[xcc].insert((se << 8) | (sh << 4) | cu); per_xcc[xcc]++;
    }
    printf("%-28s:", name);
    size_t total = 0;
    for (unsigned x : xccs) { printf(" xcc%u:%zu", x, cu_of_xcc[x].size()); total += cu_of_xcc[x].size(); }
    printf("  total CUs %zu\n", total);
    if (total <= 40) {
        for (unsigned x : xccs) { printf("    xcc%u (se.sh.cu):", x); for (unsigned c : cu_of_xcc[x]) printf(" %u.%u.%u", c >> 8, (c >> 4) & 1, c & 15); printf("\n"); }
    }
    (void)hipFree(d); (void)hipStreamDestroy(st);
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("%s: %d CUs\n", pr.name, pr.multiProcessorCount);
    auto low = [](int nbits) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < nbits; i++) m[i / 32] |= 1u << (i % 32); return m; };
    probe("all 256 bits", low(256));
    probe("low 8 bits", low(8));
    probe("low 16 bits", low(16));
    probe("low 32 bits", low(32));
    probe("low 64 bits", low(64));
    probe("low 128 bits", low(128));
    { std::vector<uint32_t> m(8, 0); for (int i = 64; i < 256; i++) m[i / 32] |= 1u << (i % 32); probe("bits 64..255", m); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); probe("every 8th bit (32 bits)", m); }
    return 0;
}
