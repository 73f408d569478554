// build: hipcc --offload-arch=gfx950 -O2 -o tools/exp/ext_event tools/exp/ext_event.hip ; run on the GPU box: ./tools/exp/ext_event
// experiment: what do hipExtLaunchKernel's start/stop events cost and measure, next to hipEventRecord markers?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float *p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x + blockIdx.x * blockDim.x] = a; }
int main()
{
    float *d; hipMalloc(&d, 1 << 24);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int N = 2000, it = 20000;          // ~ tens of us per launch
    std::vector<hipEvent_t> a(N), b(N);
    for (int i = 0; i < N; i++) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); }
    for (int mode = 0; mode < 4; mode++) {
        for (int w = 0; w < 2; w++) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, st, d, it);
                else if (mode == 1) { hipEventRecord(a[i], st); hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, st, d, it); hipEventRecord(b[i], st); }
                else if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, st, a[i], b[i], 0, d, it);
                else hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, st, nullptr, b[i], 0, d, it);
            }
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            double ev = 0; int ne = 0;
            if (mode == 1 || mode == 2) for (int i = 0; i < N; i++) { float ms; if (hipEventElapsedTime(&ms, a[i], b[i]) == hipSuccess) { ev += ms; ne++; } }
            if (mode == 3) for (int i = 0; i < N; i++) { float ms; if (hipEventElapsedTime(&ms, b[i], b[i]) == hipSuccess) { ev += ms; ne++; } }
            double gap = 0; int ng = 0;
            if (mode == 3) for (int i = 1; i < N; i++) { float ms; if (hipEventElapsedTime(&ms, b[i - 1], b[i]) == hipSuccess) { gap += ms; ng++; } }
            printf("mode %d (%s): wall %.2f us/launch, event elapsed avg %.2f us (%d ok), stop-to-stop %.2f us\n", mode,
                   mode == 0 ? "plain" : mode == 1 ? "record markers" : mode == 2 ? "ext start+stop" : "ext stop only", us, ne ? ev / ne * 1e3 : -1., ne, ng ? gap / ng * 1e3 : -1.);
        }
    }
    return 0;
}
