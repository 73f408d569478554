/* pmr_dsd.h -- C-ABI of the `dsd_in` loop body on MI355X (libpmr446_hip.so), SURVEY.md s8 row f3.
 *
 * One call = one iteration of reference src/dsd_in.c:160-178:
 *   readStream (:161, the caller's job) -> dc-block (:167) -> msresamp_crcf down to 12.5 kS/s (:168) -> freqdem (:169)
 *   -> msresamp_rrrf up to 48 kS/s (:170) -> int16 (:172-175) -> fwrite to stdout (:177, the caller's job).
 * Same conventions as pmr_chain.h: opaque handle, create returns NULL on failure (no HIP device: no CPU path), int
 * return codes with 0 == OK, caller-owned sample buffers, one thread per handle.  The output is the s16le mono
 * 48 kHz stream `dsd -i -` expects (README.md:45).
 */
#ifndef PMR_DSD_H
#define PMR_DSD_H

#include "pmr_chain.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmr_dsd_s *pmr_dsd;

typedef struct {
    double   fs_in;           /* include/dsd_in.h:11   SDR_SAMPLERATE     1024000   */
    double   sig_rate;        /* src/dsd_in.c:23       SIG_SAMPLERATE     12500     */
    double   audio_rate;      /* src/dsd_in.c:22       AUDIO_SAMPLERATE   48000     */
    float    dcblock_alpha;   /* :97   iirfilt_crcf_create_dc_blocker     0.0005    */
    float    resamp_As;       /* :100, :104  msresamp_*_create            60.0f     */
    float    fm_kf;           /* :108  freqdem_create                     0.5f      */
    unsigned max_block;       /* :25   SDR_INPUT_CHUNK                    200000    */
    int      device;          /* HIP device ordinal; -1 = current device            */
} pmr_dsd_cfg;

void     pmr_dsd_default_cfg(pmr_dsd_cfg *cfg);            /* the reference's operating point (:22-25, :97-108) */
pmr_dsd  pmr_dsd_create(const pmr_dsd_cfg *cfg);           /* init_liquid(), :95-112; NULL on failure            */
int      pmr_dsd_reset(pmr_dsd q);
int      pmr_dsd_destroy(pmr_dsd q);                       /* destroy_liquid(), :114-124                         */
unsigned pmr_dsd_max_out(pmr_dsd q);                       /* out_size rule, :140-141                            */
const char *pmr_dsd_last_error(pmr_dsd q);

/* Process one block held in HOST memory (== buffp after readStream, :161).
 *   pcm    [cap] int16, (int16_t)(x * INT16_MAX) truncated toward zero (:174), saturated; nullable
 *   audio  [cap] float32 output of the interpolator (out_buf, :170); nullable
 *   n_out  samples produced (== nz, :170); nullable.  PMR_ERANGE when n_out > cap.
 * Synchronous.                                                                                              */
int pmr_dsd_process_block(pmr_dsd q, const pmr_cf32 *iq, unsigned n_in, int16_t *pcm, float *audio, unsigned cap,
                          unsigned *n_out);
/* Device-resident variant (HIP device pointers, queued on the handle's stream, not synchronised). */
int pmr_dsd_process_block_device(pmr_dsd q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio, unsigned cap,
                                 unsigned *n_out);
int pmr_dsd_synchronize(pmr_dsd q);

/* tests: intermediates of the LAST block. what = 0: cf32 resampled stream (:168); 1: float discriminator output (:169) */
int pmr_dsd_debug_read(pmr_dsd q, int what, void *host_buf, size_t cap_bytes, size_t *n_bytes);

/* host-only helper (no HIP device needed): the closed-form sample accounting of one block */
typedef struct { uint64_t n_raw; uint64_t n_resampled; uint32_t down_phase; } pmr_dsd_plan_state;
int pmr_dsd_plan_block(const pmr_dsd_cfg *cfg, pmr_dsd_plan_state *st, unsigned n_in, unsigned *n_resampled,
                       unsigned *n_out);
/* what: 0 down stages, 1 up stages, 2 up arbitrary step, 3 down arbitrary step, 4 + g: m of up stage g */
unsigned pmr_dsd_cfg_info(const pmr_dsd_cfg *cfg, int what);

#ifdef __cplusplus
}
#endif
#endif
