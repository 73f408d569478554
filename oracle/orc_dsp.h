/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
 *
 * Scalar C restatement of the liquid-dsp v1.7.0 objects that the reference composes on its
 * per-block hot path (reference: src/sdr_pmr446.c:420-480 creates them, :795-906 runs them).
 * liquid-dsp (github.com/jgaeddert/liquid-dsp, tag v1.7.0, .github/workflows/build.yml:30-36,
 * built with --enable-simdoverride => portable scalar dot products) is NOT in /root/reference and
 * not installed; the algorithms below restate its published behaviour.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (sdr_pmr446_amd/csrc) never includes, links or calls anything in oracle/.
 */
#ifndef ORC_DSP_H
#define ORC_DSP_H

#include <complex.h>
#include <stdint.h>

typedef float complex cf32;

/* ---- filter design (liquid firdes.c / kaiser window) : SURVEY A.1 ---- */
float    orc_kaiser_beta_As(float As);
float    orc_kaiser(unsigned i, unsigned wlen, float beta);
void     orc_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h);
unsigned orc_estimate_req_filter_len(float df, float As);

/* ---- window (liquid window.proto.c): fixed-length delay line, oldest sample first ---- */
typedef struct { cf32 *buf; unsigned n, cap, pos; } orc_windowcf;
typedef struct { float *buf; unsigned n, cap, pos; } orc_windowf;
void orc_windowcf_init(orc_windowcf *w, unsigned n);
void orc_windowcf_free(orc_windowcf *w);
void orc_windowcf_reset(orc_windowcf *w);
static inline const cf32 *orc_windowcf_read(const orc_windowcf *w) { return w->buf + w->pos; }
void orc_windowcf_push(orc_windowcf *w, cf32 x);
void orc_windowf_init(orc_windowf *w, unsigned n);
void orc_windowf_free(orc_windowf *w);
void orc_windowf_reset(orc_windowf *w);
static inline const float *orc_windowf_read(const orc_windowf *w) { return w->buf + w->pos; }
void orc_windowf_push(orc_windowf *w, float x);

/* ---- dotprod (liquid dotprod.proto.c, portable scalar run4): sequential sum i = 0..n-1 ---- */
cf32  orc_dotprod_crcf(const float *h, const cf32 *x, unsigned n);
float orc_dotprod_rrrf(const float *h, const float *x, unsigned n);

/* ---- iirfilt, "norm" (direct form II) structure, any order : SURVEY A.2 ---- */
typedef struct { unsigned nb, na, n; float *b, *a; cf32 *v; } orc_iirfilt_crcf;
typedef struct { unsigned nb, na, n; float *b, *a; float *v; } orc_iirfilt_rrrf;
orc_iirfilt_crcf *orc_iirfilt_crcf_create(const float *b, unsigned nb, const float *a, unsigned na);
orc_iirfilt_crcf *orc_iirfilt_crcf_create_dc_blocker(float alpha);
void orc_iirfilt_crcf_reset(orc_iirfilt_crcf *q);
void orc_iirfilt_crcf_destroy(orc_iirfilt_crcf *q);
void orc_iirfilt_crcf_execute_block(orc_iirfilt_crcf *q, const cf32 *x, unsigned n, cf32 *y);
orc_iirfilt_rrrf *orc_iirfilt_rrrf_create(const float *b, unsigned nb, const float *a, unsigned na);
orc_iirfilt_rrrf *orc_iirfilt_rrrf_create_dc_blocker(float alpha);
void orc_iirfilt_rrrf_reset(orc_iirfilt_rrrf *q);
void orc_iirfilt_rrrf_destroy(orc_iirfilt_rrrf *q);
void orc_iirfilt_rrrf_execute_block(orc_iirfilt_rrrf *q, const float *x, unsigned n, float *y);

/* ---- resamp2_crcf half-band decimator : SURVEY A.3 ---- */
typedef struct {
    unsigned m, h_len, h1_len;
    float *h, *h1;
    orc_windowcf w0, w1;
} orc_resamp2_crcf;
orc_resamp2_crcf *orc_resamp2_crcf_create(unsigned m, float f0, float As);
void orc_resamp2_crcf_reset(orc_resamp2_crcf *q);
void orc_resamp2_crcf_destroy(orc_resamp2_crcf *q);
void orc_resamp2_crcf_decim_execute(orc_resamp2_crcf *q, const cf32 *x /*[2]*/, cf32 *y);

/* ---- msresamp2_crcf multi-stage half-band decimator ---- */
typedef struct {
    unsigned num_stages, M;
    float zeta;
    float *fc_stage, *f0_stage, *As_stage;
    unsigned *m_stage;
    orc_resamp2_crcf **stage;
    cf32 *buffer0, *buffer1;
} orc_msresamp2_crcf;
orc_msresamp2_crcf *orc_msresamp2_crcf_create_decim(unsigned num_stages, float fc, float f0, float As);
void orc_msresamp2_crcf_reset(orc_msresamp2_crcf *q);
void orc_msresamp2_crcf_destroy(orc_msresamp2_crcf *q);
void orc_msresamp2_crcf_decim_execute(orc_msresamp2_crcf *q, cf32 *x /*[M], clobbered like liquid*/, cf32 *y);

/* ---- firpfb_crcf + resamp_crcf (fixed-point phase arbitrary resampler) ---- */
typedef struct {
    unsigned m, npfb, bits_index, sub_len;
    float rate, fc, As;
    uint32_t step, phase;
    float *bank;            /* [npfb][sub_len], each sub-filter stored reversed (oldest-first order) */
    float *proto;           /* normalised prototype, 2*m*npfb+1 taps */
    orc_windowcf w;
} orc_resamp_crcf;
orc_resamp_crcf *orc_resamp_crcf_create(float rate, unsigned m, float fc, float As, unsigned npfb);
void orc_resamp_crcf_reset(orc_resamp_crcf *q);
void orc_resamp_crcf_destroy(orc_resamp_crcf *q);
void orc_resamp_crcf_execute(orc_resamp_crcf *q, cf32 x, cf32 *y, unsigned *nw);

/* ---- msresamp_crcf (decimation branch only: the path always has rate < 1) ---- */
typedef struct {
    float rate, As, rate_arbitrary, rate_halfband;
    unsigned num_halfband_stages;
    cf32 *buffer; unsigned buffer_index;
    orc_msresamp2_crcf *halfband;
    orc_resamp_crcf *arbitrary;
} orc_msresamp_crcf;
orc_msresamp_crcf *orc_msresamp_crcf_create(float rate, float As);
void orc_msresamp_crcf_reset(orc_msresamp_crcf *q);
void orc_msresamp_crcf_destroy(orc_msresamp_crcf *q);
void orc_msresamp_crcf_execute(orc_msresamp_crcf *q, const cf32 *x, unsigned nx, cf32 *y, unsigned *ny);

/* ---- msresamp_rrrf, interpolation branch (reference src/dsd_in.c:104 creates it with rate 48000/12500, :170 runs it):
 *      resamp_rrrf (arbitrary, runs first) -> msresamp2_rrrf half-band interpolators (stage 0 first) ---- */
typedef struct {
    unsigned m, h_len, h1_len;
    float *h, *h1;
    orc_windowf w0, w1;
} orc_resamp2_rrrf;
orc_resamp2_rrrf *orc_resamp2_rrrf_create(unsigned m, float f0, float As);
void orc_resamp2_rrrf_reset(orc_resamp2_rrrf *q);
void orc_resamp2_rrrf_destroy(orc_resamp2_rrrf *q);
void orc_resamp2_rrrf_interp_execute(orc_resamp2_rrrf *q, float x, float *y /*[2]*/);

typedef struct {
    unsigned m, npfb, bits_index, sub_len;
    float rate, fc, As;
    uint32_t step, phase;
    float *bank, *proto;
    orc_windowf w;
} orc_resamp_rrrf;
orc_resamp_rrrf *orc_resamp_rrrf_create(float rate, unsigned m, float fc, float As, unsigned npfb);
void orc_resamp_rrrf_reset(orc_resamp_rrrf *q);
void orc_resamp_rrrf_destroy(orc_resamp_rrrf *q);
void orc_resamp_rrrf_execute(orc_resamp_rrrf *q, float x, float *y, unsigned *nw);

typedef struct {
    float rate, As, rate_arbitrary, rate_halfband;
    unsigned num_halfband_stages;
    unsigned *m_stage;
    orc_resamp2_rrrf **stage;     /* [num_halfband_stages], stage 0 = lowest rate, runs first when interpolating */
    orc_resamp_rrrf *arbitrary;
    float *buffer0, *buffer1;
} orc_msresamp_rrrf;
orc_msresamp_rrrf *orc_msresamp_rrrf_create(float rate, float As);   /* rate >= 1 only */
void orc_msresamp_rrrf_reset(orc_msresamp_rrrf *q);
void orc_msresamp_rrrf_destroy(orc_msresamp_rrrf *q);
void orc_msresamp_rrrf_execute(orc_msresamp_rrrf *q, const float *x, unsigned nx, float *y, unsigned *ny);

/* ---- nco_crcf, LIQUID_VCO flavour : SURVEY A.4 ---- */
typedef struct { uint32_t theta, d_theta; } orc_nco_crcf;
uint32_t orc_nco_constrain(float theta);
void orc_nco_set_frequency(orc_nco_crcf *q, float dtheta);
void orc_nco_reset(orc_nco_crcf *q);
void orc_nco_sincos(const orc_nco_crcf *q, float *s, float *c);
static inline void orc_nco_step(orc_nco_crcf *q) { q->theta += q->d_theta; }
cf32 orc_nco_mix_down(const orc_nco_crcf *q, cf32 x);

/* ---- radix-2 FFT (stands in for liquid's internal FFT plan; forward, unscaled) ---- */
typedef struct { unsigned n, log2n; cf32 *tw; unsigned *rev; } orc_fft;
orc_fft *orc_fft_create(unsigned n);
void orc_fft_destroy(orc_fft *f);
void orc_fft_forward(const orc_fft *f, const cf32 *in, cf32 *out);

/* ---- firpfbch_crcf analyzer : SURVEY A.5 ---- */
typedef struct {
    unsigned M, p, h_len, filter_index;
    float *h;               /* prototype, 2*M*m+1 */
    float *dp;              /* [M][p] sub-filters, reversed */
    orc_windowcf *w;        /* [M] */
    cf32 *X, *x;
    orc_fft *fft;
} orc_firpfbch_crcf;
orc_firpfbch_crcf *orc_firpfbch_crcf_create_kaiser(unsigned M, unsigned m, float As);
void orc_firpfbch_crcf_reset(orc_firpfbch_crcf *q);
void orc_firpfbch_crcf_destroy(orc_firpfbch_crcf *q);
void orc_firpfbch_crcf_analyzer_execute(orc_firpfbch_crcf *q, const cf32 *x /*[M]*/, cf32 *y /*[M]*/);

/* ---- freqdem : SURVEY A.6 ---- */
typedef struct { float kf, ref; cf32 r_prime; } orc_freqdem;
void orc_freqdem_init(orc_freqdem *q, float kf);
static inline void orc_freqdem_reset(orc_freqdem *q) { q->r_prime = 0; }
void orc_freqdem_demodulate_block(orc_freqdem *q, const cf32 *r, unsigned n, float *m);

/* ---- firfilt_rrrf : SURVEY A.7 ---- */
typedef struct { unsigned n; float *hr; /* reversed taps */ float scale; orc_windowf w; } orc_firfilt_rrrf;
orc_firfilt_rrrf *orc_firfilt_rrrf_create(const float *h, unsigned n);
void orc_firfilt_rrrf_reset(orc_firfilt_rrrf *q);
void orc_firfilt_rrrf_destroy(orc_firfilt_rrrf *q);
void orc_firfilt_rrrf_execute_block(orc_firfilt_rrrf *q, const float *x, unsigned n, float *y);

/* ---- wdelayf : SURVEY A.8 ---- */
typedef struct { unsigned delay, idx; float *v; } orc_wdelayf;
orc_wdelayf *orc_wdelayf_create(unsigned delay);
void orc_wdelayf_reset(orc_wdelayf *q);
void orc_wdelayf_destroy(orc_wdelayf *q);
static inline void orc_wdelayf_push(orc_wdelayf *q, float x) {
    q->v[q->idx] = x; q->idx = (q->idx + 1) % (q->delay + 1);
}
static inline float orc_wdelayf_read(const orc_wdelayf *q) { return q->v[q->idx]; }

/* ---- cbuffercf : SURVEY A.9 ---- */
typedef struct { cf32 *v; unsigned max_size, max_read, num_alloc, num, ri, wi; } orc_cbuffercf;
orc_cbuffercf *orc_cbuffercf_create(unsigned max_size);
void orc_cbuffercf_reset(orc_cbuffercf *q);
void orc_cbuffercf_destroy(orc_cbuffercf *q);
static inline unsigned orc_cbuffercf_size(const orc_cbuffercf *q) { return q->num; }
int  orc_cbuffercf_write(orc_cbuffercf *q, const cf32 *v, unsigned n);
void orc_cbuffercf_read(orc_cbuffercf *q, unsigned n, cf32 **v, unsigned *nread);
int  orc_cbuffercf_release(orc_cbuffercf *q, unsigned n);

/* ---- spgramcf / asgramcf : the waterfall line of the reference (src/sdr_pmr446.c:473-477 create + set_scale(-40, 2),
 * :911-912 write(resamp_buf, ny) + execute per block).  liquid-dsp v1.7.0 src/fft/src/spgram.proto.c, asgram.proto.c restated
 * from knowledge of the library [structure M, window normalisation L]; where confidence < H this restatement DEFINES the
 * behaviour (SURVEY App. A rule):
 *   asgram(nfft): display width nfft, transform size nfftp = 4 nfft, spgram(nfftp, HANN, window_len = nfft, delay = nfft / 2);
 *   spgram: every `delay` pushed samples, FFT of (last window_len samples) * w, zero-padded to nfftp; psd += |X|^2;
 *           w[i] = hann(i, window_len) * sqrt(2) / (sqrt(sum w^2 / window_len) * sqrt(nfftp));
 *           get_psd: 10 log10(max(1e-12, psd[(i + nfftp/2) % nfftp] / num_transforms));
 *   asgram execute: get_psd, spgram reset (accumulators AND window buffer: every call starts from an empty window), peak over the
 *           nfftp bins, one character per group of 4 bins (the group's maximum against levels ref + k div, " .,-+*&NM#"). ---- */
typedef struct {
    unsigned nfft, window_len, delay, sample_timer;
    unsigned long long num_transforms;
    float *w; cf32 *buf_time, *buf_freq; float *psd;
    orc_windowcf buffer; orc_fft *fft;
} orc_spgramcf;
orc_spgramcf *orc_spgramcf_create(unsigned nfft, unsigned window_len, unsigned delay);   /* Hann window; nfft a power of two */
void orc_spgramcf_destroy(orc_spgramcf *q);
void orc_spgramcf_reset(orc_spgramcf *q);
void orc_spgramcf_write(orc_spgramcf *q, const cf32 *x, unsigned n);
void orc_spgramcf_get_psd(const orc_spgramcf *q, float *psd_db /*[nfft]*/);

typedef struct { unsigned nfft, nfftp, p; orc_spgramcf *periodogram; float *psd; float levels[10]; char levelchar[10];
                 unsigned num_levels; float div, ref; } orc_asgramcf;
orc_asgramcf *orc_asgramcf_create(unsigned nfft);
void orc_asgramcf_destroy(orc_asgramcf *q);
void orc_asgramcf_set_scale(orc_asgramcf *q, float ref, float div);
void orc_asgramcf_write(orc_asgramcf *q, const cf32 *x, unsigned n);
/* ascii: nfft characters (no terminator); psd_db_out (nullable): the nfftp PSD values the characters were drawn from */
void orc_asgramcf_execute(orc_asgramcf *q, char *ascii, float *peakval, float *peakfreq, float *psd_db_out);

#endif
