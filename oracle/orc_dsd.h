/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
 *
 * CPU restatement of the loop body of the reference's second executable, `dsd_in`
 * (src/dsd_in.c:160-178): dc-block -> msresamp_crcf down to 12.5 kS/s -> freqdem -> msresamp_rrrf up to 48 kS/s ->
 * int16 (the s16le mono 48 kHz wire format piped into `dsd`, README.md:45).  SURVEY.md s8 row f3.
 */
#ifndef ORC_DSD_H
#define ORC_DSD_H

#include <stdint.h>
#include "orc_dsp.h"

typedef struct {
    double   fs_in;           /* include/dsd_in.h:11   SDR_SAMPLERATE 1024000            */
    double   sig_rate;        /* src/dsd_in.c:23       SIG_SAMPLERATE 12500              */
    double   audio_rate;      /* src/dsd_in.c:22       AUDIO_SAMPLERATE 48000            */
    float    dcblock_alpha;   /* :97                   0.0005                            */
    float    resamp_As;       /* :100,:104             60.0f                             */
    float    fm_kf;           /* :108                  0.5f                              */
    unsigned max_block;       /* :25                   SDR_INPUT_CHUNK 200000            */
} orc_dsd_cfg;

typedef struct {
    orc_dsd_cfg cfg;
    unsigned res_size, out_size;             /* :140-141 */
    orc_iirfilt_crcf  *dcblock;
    orc_msresamp_crcf *res_down;
    orc_msresamp_rrrf *res_up;
    orc_freqdem        fm_demod;
    cf32 *buffp, *resamp_buf; float *fm_out_buf, *out_buf;
} orc_dsd;

void     orc_dsd_default_cfg(orc_dsd_cfg *cfg);
orc_dsd *orc_dsd_create(const orc_dsd_cfg *cfg);
int      orc_dsd_reset(orc_dsd *q);
int      orc_dsd_destroy(orc_dsd *q);
unsigned orc_dsd_max_out(const orc_dsd *q);          /* == out_size, :141 */
unsigned orc_dsd_max_resampled(const orc_dsd *q);    /* == res_size, :140 */
/* what: 0 down stages, 1 up stages, 2 up arbitrary step, 3 down arbitrary step, 4 + g: m of up stage g */
unsigned orc_dsd_info(const orc_dsd *q, int what);

/* designed coefficients: what 0 = up arbitrary bank [256][14] (oldest-first), 1 + g = branch taps of up stage g */
unsigned orc_dsd_design(const orc_dsd *q, int what, float *out, unsigned cap);

/* One loop iteration (:167-178).  pcm / audio (nullable) receive n_out samples (cap >= orc_dsd_max_out);
 * resampled (nullable, cf32[res_size]) and fm (nullable, float[res_size]) are tap-offs of :168 / :169. */
int orc_dsd_process_block(orc_dsd *q, const cf32 *iq, unsigned n_in, int16_t *pcm, float *audio, unsigned cap,
                          unsigned *n_out, cf32 *resampled, float *fm, unsigned *n_resampled);

#endif
