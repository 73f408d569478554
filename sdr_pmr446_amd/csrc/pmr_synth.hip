// pmr_synth.hip -- include/pmr_mem.h: device / pinned memory helpers and the synthetic multi-channel NBFM test signal
//  generated directly in HBM.  Stands in for the SoapySDR cf32 ingest (reference src/shared.c:62,
// src/sdr_pmr446.c:789); channel plan of sdr_pmr446_amd/synth.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/pmr_mem.h"
#include "../data/pmr446_taps.h"          /* pmr446_ctcss_freqs (reference :138-141) */
#include "pmr_kernels.h"                  /* PMR_KLAUNCH, poison mode */

extern "C" void *pmr_device_alloc(size_t bytes, int device)
{
    if (device < -1) return NULL;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return NULL; }    /* (the error is reported, not left sticky) */
    void *p = NULL;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return NULL;
    /* zero-filled; 0xFF bytes in the test-only poison mode (an output row the library was to write and did not then shows) */
    if (hipMemset(p, pmr_debug_poison_enabled() ? 0xFF : 0, bytes ? bytes : 16) != hipSuccess) { (void)hipFree(p); return NULL; }
    return p;
}
extern "C" void pmr_device_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int pmr_memcpy_h2d(void *d, const void *h, size_t n) { return n ? (int)hipMemcpy(d, h, n, hipMemcpyHostToDevice) : 0; }
extern "C" int pmr_memcpy_d2h(void *h, const void *d, size_t n) { return n ? (int)hipMemcpy(h, d, n, hipMemcpyDeviceToHost) : 0; }
extern "C" int pmr_device_synchronize(void) { return (int)hipDeviceSynchronize(); }

extern "C" void pmr_synth_default_cfg(pmr_synth_cfg *c, double fs_in, unsigned num_channels)
{
    c->fs_in = fs_in; c->num_channels = num_channels; c->stream_id = 0; c->snr_db = 30.0; c->dev_hz = 2500.0;
    c->ctcss_dev_hz = 300.0; c->period_log2 = 0; c->channel_step = 1;
}

// per-channel oscillator parameters: phases are 32-bit fixed point in revolutions (exactly continuous for any n0)
struct synth_chan { uint32_t inc_c, ph0, inc_a, inc_t; float beta_a, beta_t; };

static __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_synth_iq(float2 *__restrict__ out, uint64_t n0, size_t n, const synth_chan *__restrict__ ch,
                                                  unsigned nch, float amp, float sigma, uint64_t seed)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = n0 + i;
    const uint32_t a32 = (uint32_t)a;
    float re = 0.f, im = 0.f;
    const float rev = 1.0f / 4294967296.0f;
    for (unsigned k = 0; k < nch; k++) {                                   // wave-uniform table walk (scalar loads)
        const synth_chan c = ch[k];
        const float pa = (float)(a32 * c.inc_a) * rev, pt = (float)(a32 * c.inc_t) * rev;
        // theta / 2 pi = carrier revolutions + (beta_a sin(2 pi pa) + beta_t sin(2 pi pt)) / 2 pi
        const float fm = c.beta_a * __builtin_amdgcn_sinf(pa) + c.beta_t * __builtin_amdgcn_sinf(pt);
        const float th = (float)(a32 * c.inc_c + c.ph0) * rev + fm;
        const float fr = th - floorf(th);
        re += __builtin_amdgcn_cosf(fr);
        im += __builtin_amdgcn_sinf(fr);
    }
    // complex AWGN, counter-based (any sub-range of the stream is reproducible)
    const uint64_t z1 = splitmix64(seed + 2 * a), z2 = splitmix64(seed + 2 * a + 1);
    const float u1 = ((float)(z1 >> 40) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(z2 >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-logf(u1)) * sigma;
    out[i] = make_float2(fmaf(amp, re, r * __builtin_amdgcn_cosf(u2)), fmaf(amp, im, r * __builtin_amdgcn_sinf(u2)));
}

extern "C" int pmr_synth_iq_device(const pmr_synth_cfg *c, void *d_out, uint64_t n0, size_t n)
{
    if (!c || !d_out || !c->num_channels || !(c->fs_in > 0)) return 1;
    if (!n) return 0;
    const unsigned M = c->num_channels, step = c->channel_step ? c->channel_step : 1;
    const double fs = c->fs_in, W = 12500.0;
    const uint64_t seed = 0x504D523434343600ull + c->stream_id;
    // frequency -> 32-bit phase increment; with period_log2 = b the increment is a multiple of 2^(32-b): 2^b samples = whole cycles
    const unsigned b = c->period_log2 > 32 ? 32 : c->period_log2;
    const double q = b ? ldexp(1.0, 32 - (int)b) : 1.0;
    const auto inc = [&](double f) { return (uint32_t)(int64_t)llround(llround(f / fs * 4294967296.0 / q) * q); };
    synth_chan *tab = (synth_chan *)calloc(M, sizeof(synth_chan));
    if (!tab) return 1;
    unsigned nch = 0;
    uint64_t ps = seed ^ 0xA5A5A5A5ull;
    for (unsigned k = 0; k < M; k += step) {
        if (k % 8 == 7) continue;                                          // empty channel (noise only)
        synth_chan s;
        s.inc_c = inc(((double)k - (M - 1) / 2.0) * W);
        uint64_t z = ps + k;                                               // per-channel start phase
        z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        s.ph0 = (uint32_t)(z >> 32);
        const int fm = k % 8 != 3;                                         // k % 8 == 3: bare carrier
        const double fa = 400.0 + 37.0 * (k % 64), ft = (double)pmr446_ctcss_freqs[k % 38];
        s.inc_a = inc(fa); s.inc_t = inc(ft);
        const double fa_e = (double)s.inc_a / 4294967296.0 * fs, ft_e = (double)s.inc_t / 4294967296.0 * fs;
        s.beta_a = fm && fa_e > 0 ? (float)(c->dev_hz / fa_e / (2.0 * M_PI)) : 0.f;         // in revolutions
        s.beta_t = fm && ft_e > 0 ? (float)(c->ctcss_dev_hz / ft_e / (2.0 * M_PI)) : 0.f;
        tab[nch++] = s;
    }
    synth_chan *d_tab = NULL;
    hipError_t e = hipMalloc((void **)&d_tab, (nch ? nch : 1) * sizeof(synth_chan));
    if (e == hipSuccess && nch) e = hipMemcpy(d_tab, tab, nch * sizeof(synth_chan), hipMemcpyHostToDevice);
    free(tab);
    if (e != hipSuccess) { if (d_tab) (void)hipFree(d_tab); return (int)e; }
    const float amp = (float)(0.5 / sqrt((double)M));
    const float sigma = (float)sqrt((double)amp * amp / pow(10.0, c->snr_db / 10.0) * (fs / W));
    const size_t chunk = (size_t)1 << 24;                                  // bounded launches
    for (size_t p0 = 0; p0 < n && e == hipSuccess; p0 += chunk) {
        const size_t m = n - p0 < chunk ? n - p0 : chunk;
        PMR_KLAUNCH(k_synth_iq, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, (float2 *)d_out + p0, n0 + p0, m, d_tab, nch,
                           amp, sigma, seed);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_tab);
    return (int)e;
}
