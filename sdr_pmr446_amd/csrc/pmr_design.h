/* pmr_design.h -- host-side filter/phase design for the chain (what init_liquid() fixes,
 * reference src/sdr_pmr446.c:420-480).  Plain C; no device code here. */
#ifndef PMR_DESIGN_H
#define PMR_DESIGN_H

#include <stdint.h>

#define PMR_MAX_STAGES 24
#define PMR_ARB_M      7       /* resamp_crcf semi-length used by msresamp */
#define PMR_ARB_NPFB   256
#define PMR_ARB_BITS   8

typedef struct {
    /* rates (src/sdr_pmr446.c:27,425-426) */
    float    rate;              /* (M*channel_width)/fs_in as float                         */
    float    rate_arb;          /* rate * 2^num_stages in [0.5,1)                            */
    unsigned num_stages;        /* half-band decimation stages                               */
    unsigned decim;             /* 2^num_stages                                              */
    float    zeta;              /* 1/decim, applied once at the cascade output               */
    /* half-band stages, design index g (stage num_stages-1 executes first)                  */
    unsigned m_stage[PMR_MAX_STAGES];
    float   *hb_proto[PMR_MAX_STAGES];   /* 4m+1 prototype                                   */
    float   *hb_h1[PMR_MAX_STAGES];      /* 2m branch taps, oldest-sample-first order        */
    /* arbitrary resampler                                                                   */
    uint32_t arb_step;          /* round(2^24 / rate_arb)                                    */
    float   *arb_proto;         /* 2*m*npfb+1, normalised so sum == npfb                     */
    float   *arb_bank;          /* [npfb][2m] oldest-sample-first                            */
    /* NCO (:430-434)                                                                        */
    uint32_t nco_dtheta;
    unsigned nco_period;        /* 2^32 / gcd(dtheta, 2^32), 0 if > PMR_NCO_MAX_PERIOD       */
    float   *nco_cs;            /* [period][2] = (cos, sin) of phase k*dtheta                */
    /* channelizer (:436-437)                                                                */
    unsigned M, pfb_m, pfb_p;
    float   *pfb_proto;         /* 2*M*m+1                                                   */
    float   *pfb_taps_t;        /* [p][M]: element [k][c] multiplies frame (t-(p-1)+k), input phase c:
                                   = proto[(M-1-c) + (p-1-k)*M]  (oldest-first accumulation)  */
    float   *fft_tw;            /* [M/2][2] = (cos, sin) of -2*pi*k/M                        */
    /* dc blocker (:422)                                                                     */
    float    dc_a1;             /* -1 + alpha (float)                                        */
    double   dc_lambda;         /* -dc_a1 as double                                          */
    /* discriminator (:440)                                                                  */
    float    fm_ref;            /* 1/(2*pi*kf)                                               */
    /* de-emphasis IIR (:461-463)                                                            */
    float    de_b0, de_b1, de_a1;
} pmr_design;

#define PMR_NCO_MAX_PERIOD 8192

int  pmr_design_build(pmr_design *d, double fs_in, unsigned M, double channel_width_hz, float dc_alpha,
                      float resamp_As, unsigned pfb_m, float pfb_As, float fm_kf);
void pmr_design_free(pmr_design *d);

/* msresamp_rrrf interpolator of `dsd_in` (reference src/dsd_in.c:104): arbitrary resampler first, then half-band
 * interpolators, design stage 0 (lowest rate) first */
#define PMR_UP_MAX_STAGES 8
typedef struct {
    float    rate, rate_arb;    /* requested rate; rate / 2^num_stages in (1, 2]               */
    unsigned num_stages;
    unsigned m_stage[PMR_UP_MAX_STAGES];
    float   *hb_h1[PMR_UP_MAX_STAGES];   /* 2m branch taps, oldest-sample-first                 */
    uint32_t arb_step;          /* round(2^24 / rate_arb)                                       */
    float   *arb_bank;          /* [npfb][2m] oldest-sample-first                               */
} pmr_up_design;
int  pmr_up_design_build(pmr_up_design *u, float rate, float As);
void pmr_up_design_free(pmr_up_design *u);

/* sizing rule of src/sdr_pmr446.c:730-736 */
void pmr_design_buffer_sizes(const pmr_design *d, unsigned max_block, unsigned *res_size, unsigned *chan_size);

#endif
