/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
 *
 * CPU restatement of the reference's per-block loop body (src/sdr_pmr446.c:795-906) composed from
 * the restated liquid objects in orc_dsp.[ch], generalised as BASELINE.json's north_star asks:
 * runtime M / fs_in, and NBFM demodulation of ALL M channels (the reference demodulates only the
 * squelch-selected channel, :876-877; `only_channel` >= 0 restores that behaviour).
 */
#ifndef ORC_CHAIN_H
#define ORC_CHAIN_H

#include <stdint.h>
#include "orc_dsp.h"

typedef struct {
    double   fs_in;              /* include/sdr_pmr446.h:13  SDR_SAMPLERATE                 */
    unsigned num_channels;       /* src/sdr_pmr446.c:23      NUM_CHANNELS                   */
    double   channel_width_hz;   /* :22                      CHANNEL_WIDTH_HZ               */
    float    dcblock_alpha;      /* :422                     0.0005f                        */
    float    resamp_As;          /* :426                     60.0f                          */
    unsigned pfb_m;              /* :437                     13                             */
    float    pfb_As;             /* :437                     80.0                           */
    float    fm_kf;              /* :440                     0.5f                           */
    float    audio_gain;         /* :33,:890                 4.0                            */
    int      lowpass;            /* :154,:900                0                              */
    int      deemph_fir;         /* :457  APP_FIR_DEEMPH     0 = IIR                        */
    unsigned max_block;          /* :30   SDR_INPUT_CHUNK    100000                         */
    int      only_channel;       /* -1 = demodulate all channels; k = reference semantics   */
    unsigned ctcss_block;        /* :46   CTCSS_BLOCK_SIZE   2441 (Goertzel length)         */
    const float *hp_taps; unsigned hp_len;          /* :56-104, 377  */
    const float *lp_taps; unsigned lp_len;          /* :106-119, 103 */
    const float *deemph_taps; unsigned deemph_len;  /* :121-136, 101 */
} orc_chain_cfg;

/* CTCSS tone detector (reference src/sdr_pmr446.c:338-418, struct include/sdr_pmr446.h:42-52) */
#define ORC_CTCSS_NUM_FREQS 38
typedef struct {
    float coef[ORC_CTCSS_NUM_FREQS], u0[ORC_CTCSS_NUM_FREQS], u1[ORC_CTCSS_NUM_FREQS], power[ORC_CTCSS_NUM_FREQS];
    float max_power; int max_power_index; unsigned samp_processed; int tone_detected;
} orc_ctcss_detector;
typedef struct { int index; int detected; float max_power; float avg_power; } orc_ctcss_event;

typedef struct {
    orc_freqdem        fm_demod;
    orc_firfilt_rrrf  *ctcss_filt;
    orc_wdelayf       *ctcss_lp_delay;
    orc_firfilt_rrrf  *audio_filt;
    orc_iirfilt_rrrf  *deemph_iir;
    orc_firfilt_rrrf  *deemph_fir;
    orc_iirfilt_rrrf  *ctcss_dcblock;    /* :450, run :606 */
    orc_ctcss_detector ctcss;            /* :777 */
} orc_chan_state;

typedef struct orc_chain_s {
    orc_chain_cfg cfg;
    unsigned M, res_size, chan_size;
    orc_iirfilt_crcf  *dcblock;
    orc_msresamp_crcf *resampler;
    orc_nco_crcf       nco;
    orc_firpfbch_crcf *channelizer;
    orc_cbuffercf     *resamp_ring;
    orc_chan_state    *ch;          /* [M] */
    cf32 *buffp, *resamp_buf, *tmp_chan_out, *chan_bufs;  /* chan_bufs [M][chan_size] */
    float *tmp1, *tmp2;
} orc_chain;

/* optional tap-offs (any pointer may be NULL) */
typedef struct {
    cf32    *resampled;   unsigned resampled_cap; unsigned n_resampled;  /* output of :796          */
    float   *fm;          /* [M][stride] discriminator output (:881)                                */
    float   *ctcss_lp;    /* [M][stride] delayed - highpassed branch (:889), pre ctcss_execute      */
    float   *audio;       /* [M][stride] float audio handed to the sink (:904)                       */
    unsigned stride;
    /* CTCSS decisions of every Goertzel block completed in this call (:381-406): [M][ctcss_cap] */
    orc_ctcss_event *ctcss_events; unsigned ctcss_cap; unsigned ctcss_n;
} orc_taps;

void       orc_chain_default_cfg(orc_chain_cfg *cfg);
orc_chain *orc_chain_create(const orc_chain_cfg *cfg);
int        orc_chain_reset(orc_chain *q);
int        orc_chain_seek(orc_chain *q, uint64_t n_raw);               /* test hook: the state after n_raw zero samples */
unsigned   orc_ctcss_detector_run(const float *xs, unsigned nx, double audio_rate, unsigned block, orc_ctcss_event *ev, unsigned cap,
                                  float *powers /*nullable [cap][38]*/);
void       orc_deemph_iir_coefs(float b[2], float a[2]);
void       orc_dcblock_rrrf_run(const float *x, unsigned n, float alpha, float *y);
int        orc_chain_reset_channel(orc_chain *q, unsigned channel);   /* freqdem_reset + ctcss_detector_reset, :866-867 */
int        orc_chain_destroy(orc_chain *q);
unsigned   orc_chain_max_frames(const orc_chain *q);
unsigned   orc_chain_max_resampled(const orc_chain *q);

/* One reference loop iteration.  pcm/chan_out are channel-major [M][stride]; rssi_db is [M].
 * Returns 0 on success (liquid's LIQUID_OK convention), non-zero on error.                        */
int orc_chain_process_block(orc_chain *q, const cf32 *iq, unsigned n_in,
                            int16_t *pcm, unsigned pcm_stride, unsigned *n_frames,
                            cf32 *chan_out, float *rssi_db, orc_taps *taps);

int16_t orc_pcm_from_float(float x);
unsigned orc_chain_info(const orc_chain *q, int what, unsigned idx);
unsigned orc_chain_design(const orc_chain *q, int what, unsigned idx, float *out, unsigned cap);

#endif
