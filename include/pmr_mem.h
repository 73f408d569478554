/* pmr_mem.h -- device / pinned-host memory and the synthetic test signal, from THIS library's HIP runtime.
 *
 * A caller that keeps its IQ blocks in HBM (the device entry point of pmr_chain.h) needs some way to put them there.  A C
 * caller has hipMalloc / hipMemcpy; a Python caller that is NOT also a PyTorch program should not have to import one (and a
 * PyTorch wheel bundles its own copy of the HIP runtime, a second one beside the system runtime this library links: pointers
 * shared between the two work only by accident of initialisation order).  These calls are that seam: bench.py and the C
 * harness allocate, fill and read back every buffer through them.
 *
 * pmr_synth_iq_device is the synthetic ingest of SURVEY.md s8(d), standing in for SoapySDR's readStream (reference
 * src/shared.c:62, src/sdr_pmr446.c:789): M NBFM channels on the PMR446 raster + AWGN, generated directly in HBM by a kernel.
 * Same channel plan as sdr_pmr446_amd/synth.py (the numpy version the parity tests feed to both implementations), not the
 * same noise realisation.
 */
#ifndef PMR_MEM_H
#define PMR_MEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void *pmr_device_alloc(size_t bytes, int device /* -1 = current */);     /* zero-filled; NULL on failure */
void  pmr_device_free(void *p);
int   pmr_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);       /* synchronous; 0 == OK */
int   pmr_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);
int   pmr_device_synchronize(void);

typedef struct {
    double   fs_in;              /* input sample rate                                                          */
    unsigned num_channels;       /* M: channel k sits at (k - (M-1)/2) * 12.5 kHz from band centre (:25-28)    */
    unsigned stream_id;          /* seed = 0x504D523434343600 + stream_id                                      */
    double   snr_db;             /* per-channel SNR in 12.5 kHz (30)                                           */
    double   dev_hz;             /* peak deviation of the audio tone 400 + 37 (k mod 64) Hz (2500)             */
    double   ctcss_dev_hz;       /* deviation of the CTCSS tone ctcss_freqs[k mod 38] (300)                    */
    unsigned period_log2;        /* 0: free-running; b: every frequency snapped so that a block of 2^b samples
                                    repeated back to back is one phase-continuous stream                       */
    unsigned channel_step;       /* synthesise channels 0, step, 2 step, ... only (1 = all); amplitudes and noise
                                    do not depend on it                                                        */
} pmr_synth_cfg;
void pmr_synth_default_cfg(pmr_synth_cfg *c, double fs_in, unsigned num_channels);
/* samples [n0, n0 + n) of the stream into d_out (cf32, device memory); queued on the null stream and synchronised */
int  pmr_synth_iq_device(const pmr_synth_cfg *c, void *d_out, uint64_t n0, size_t n);

#ifdef __cplusplus
}
#endif
#endif
