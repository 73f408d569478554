/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md and orc_dsp.h).
 *
 * Each function cites (a) the reference call site in /root/reference that uses the liquid-dsp
 * object being restated and (b) the SURVEY.md Appendix A paragraph that records the algorithm.
 * All sample arithmetic is IEEE float32, products and sums in the order liquid's portable
 * (--enable-simdoverride) code performs them; compile with -ffp-contract=off.
 * Filter DESIGN is evaluated in double and rounded once to float32 (liquid evaluates in float;
 * the difference is a last-bit tap perturbation -- a documented choice of this restatement).
 */
#include "orc_dsp.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* filter design: liquid firdes.c, used by firpfbch_crcf_create_kaiser (ref src/sdr_pmr446.c:436)
 * and inside msresamp_crcf_create (ref :425).  SURVEY A.1.                                    */

float orc_kaiser_beta_As(float As)
{
    As = fabsf(As);
    float beta;
    if (As > 50.0f)      beta = 0.1102f * (As - 8.7f);
    else if (As > 21.0f) beta = 0.5842f * powf(As - 21.0f, 0.4f) + 0.07886f * (As - 21.0f);
    else                 beta = 0.0f;
    return beta;
}

static double besseli0(double z)
{
    /* power series sum_k ((z/2)^k / k!)^2 -- converges quickly for z <= ~10 used here */
    double y = 1.0, t = 1.0, hz = 0.5 * z;
    for (unsigned k = 1; k < 200; k++) {
        t *= hz / (double)k;
        double tt = t * t;
        y += tt;
        if (tt < 1e-20 * y) break;
    }
    return y;
}

float orc_kaiser(unsigned i, unsigned wlen, float beta)
{
    double t = (double)i - (double)(wlen - 1) / 2.0;
    double r = 2.0 * t / (double)(wlen - 1);
    double a = 1.0 - r * r;
    if (a < 0.0) a = 0.0;
    return (float)(besseli0((double)beta * sqrt(a)) / besseli0((double)beta));
}

static double sinc_d(double x)
{
    if (fabs(x) < 1e-12) return 1.0;
    return sin(M_PI * x) / (M_PI * x);
}

void orc_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h)
{
    float beta = orc_kaiser_beta_As(As);
    double ib = besseli0((double)beta);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0 + (double)mu;
        double r = 2.0 * ((double)i - (double)(n - 1) / 2.0) / (double)(n - 1);
        double a = 1.0 - r * r;
        if (a < 0.0) a = 0.0;
        double w = besseli0((double)beta * sqrt(a)) / ib;
        h[i] = (float)(sinc_d(2.0 * (double)fc * t) * w);
    }
}

unsigned orc_estimate_req_filter_len(float df, float As)
{
    /* Kaiser's estimate, truncated to unsigned like liquid's estimate_req_filter_len() */
    float h_len = (As - 7.95f) / (14.26f * df);
    return (unsigned)h_len;
}

/* ------------------------------------------------------------------------------------------ */
/* window: liquid window.proto.c -- push newest at the end, read pointer exposes n samples,
 * oldest first.  Linearised buffer with periodic memmove (same idea as liquid).               */

#define ORC_WIN_IMPL(NAME, T)                                                              \
    void NAME##_init(NAME *w, unsigned n) {                                                \
        w->n = n; w->cap = 4 * n + 64; w->pos = 0;                                         \
        w->buf = (T *)calloc(w->cap, sizeof(T));                                           \
    }                                                                                      \
    void NAME##_free(NAME *w) { free(w->buf); w->buf = NULL; }                             \
    void NAME##_reset(NAME *w) { memset(w->buf, 0, w->cap * sizeof(T)); w->pos = 0; }      \
    void NAME##_push(NAME *w, T x) {                                                       \
        if (w->pos + w->n == w->cap) {                                                     \
            memmove(w->buf, w->buf + w->pos + 1, (w->n - 1) * sizeof(T));                  \
            w->pos = 0;                                                                    \
        } else {                                                                           \
            w->pos++;                                                                      \
        }                                                                                  \
        w->buf[w->pos + w->n - 1] = x;                                                     \
    }
ORC_WIN_IMPL(orc_windowcf, cf32)
ORC_WIN_IMPL(orc_windowf, float)

/* ------------------------------------------------------------------------------------------ */
/* dotprod: liquid dotprod.proto.c DOTPROD(_run4): r += h[i]*x[i] for i = 0..n-1 in order.
 * crcf: real coefficient times complex sample = (h*re, h*im).  SURVEY A.7 (summation order).  */

cf32 orc_dotprod_crcf(const float *h, const cf32 *x, unsigned n)
{
    float re = 0.0f, im = 0.0f;
    for (unsigned i = 0; i < n; i++) {
        re += h[i] * crealf(x[i]);
        im += h[i] * cimagf(x[i]);
    }
    return CMPLXF(re, im);
}

float orc_dotprod_rrrf(const float *h, const float *x, unsigned n)
{
    float r = 0.0f;
    for (unsigned i = 0; i < n; i++) r += h[i] * x[i];
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* iirfilt "norm" form: liquid iirfilt.proto.c IIRFILT(_execute_norm).  SURVEY A.2.
 * ref: iirfilt_crcf_create_dc_blocker(0.0005f) src/sdr_pmr446.c:422, run :795;
 *      iirfilt_rrrf_create(b,2,a,2) :461-463, run :898; iirfilt_rrrf dc blocker :450, run :606. */

#define ORC_IIR_IMPL(NAME, T)                                                              \
    NAME *NAME##_create(const float *b, unsigned nb, const float *a, unsigned na) {        \
        NAME *q = (NAME *)calloc(1, sizeof(NAME));                                         \
        q->nb = nb; q->na = na; q->n = nb > na ? nb : na;                                  \
        q->b = (float *)calloc(q->nb, sizeof(float));                                      \
        q->a = (float *)calloc(q->na, sizeof(float));                                      \
        float a0 = a[0];                                                                   \
        for (unsigned i = 0; i < nb; i++) q->b[i] = b[i] / a0;                             \
        for (unsigned i = 0; i < na; i++) q->a[i] = a[i] / a0;                             \
        q->v = (T *)calloc(q->n, sizeof(T));                                               \
        return q;                                                                          \
    }                                                                                      \
    NAME *NAME##_create_dc_blocker(float alpha) {                                          \
        float a1 = -1.0f + alpha;                                                          \
        float b[2] = {1.0f, -1.0f};                                                        \
        float a[2] = {1.0f, a1};                                                           \
        return NAME##_create(b, 2, a, 2);                                                  \
    }                                                                                      \
    void NAME##_reset(NAME *q) { memset(q->v, 0, q->n * sizeof(T)); }                      \
    void NAME##_destroy(NAME *q) { if (!q) return; free(q->b); free(q->a); free(q->v); free(q); }

ORC_IIR_IMPL(orc_iirfilt_crcf, cf32)
ORC_IIR_IMPL(orc_iirfilt_rrrf, float)

void orc_iirfilt_crcf_execute_block(orc_iirfilt_crcf *q, const cf32 *x, unsigned n, cf32 *y)
{
    for (unsigned k = 0; k < n; k++) {
        for (unsigned i = q->n - 1; i > 0; i--) q->v[i] = q->v[i - 1];
        float v0r = crealf(x[k]), v0i = cimagf(x[k]);
        for (unsigned i = 1; i < q->na; i++) {
            v0r -= q->a[i] * crealf(q->v[i]);
            v0i -= q->a[i] * cimagf(q->v[i]);
        }
        q->v[0] = CMPLXF(v0r, v0i);
        float yr = 0.0f, yi = 0.0f;
        for (unsigned i = 0; i < q->nb; i++) {
            yr += q->b[i] * crealf(q->v[i]);
            yi += q->b[i] * cimagf(q->v[i]);
        }
        y[k] = CMPLXF(yr, yi);
    }
}

void orc_iirfilt_rrrf_execute_block(orc_iirfilt_rrrf *q, const float *x, unsigned n, float *y)
{
    for (unsigned k = 0; k < n; k++) {
        for (unsigned i = q->n - 1; i > 0; i--) q->v[i] = q->v[i - 1];
        float v0 = x[k];
        for (unsigned i = 1; i < q->na; i++) v0 -= q->a[i] * q->v[i];
        q->v[0] = v0;
        float y0 = 0.0f;
        for (unsigned i = 0; i < q->nb; i++) y0 += q->b[i] * q->v[i];
        y[k] = y0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* resamp2_crcf: liquid resamp2.proto.c.  SURVEY A.3.  Prototype length 4m+1,
 * h[i] = sinc(t/2) * kaiser(i; beta(As)) (f0 = 0 on this path); odd taps reversed form the
 * 2m-tap branch filter, the even branch is the centre-tap delay.                               */

orc_resamp2_crcf *orc_resamp2_crcf_create(unsigned m, float f0, float As)
{
    orc_resamp2_crcf *q = (orc_resamp2_crcf *)calloc(1, sizeof(*q));
    q->m = m;
    q->h_len = 4 * m + 1;
    q->h1_len = 2 * m;
    q->h = (float *)calloc(q->h_len, sizeof(float));
    q->h1 = (float *)calloc(q->h1_len, sizeof(float));
    float beta = orc_kaiser_beta_As(As);
    double ib = besseli0((double)beta);
    for (unsigned i = 0; i < q->h_len; i++) {
        double t = (double)i - (double)(q->h_len - 1) / 2.0;
        double r = 2.0 * t / (double)(q->h_len - 1);
        double a = 1.0 - r * r;
        if (a < 0.0) a = 0.0;
        double h1 = sinc_d(t / 2.0);
        double h2 = besseli0((double)beta * sqrt(a)) / ib;
        double h3 = cos(2.0 * M_PI * t * (double)f0);
        q->h[i] = (float)(h1 * h2 * h3);
    }
    unsigned j = 0;
    for (unsigned i = 1; i < q->h_len; i += 2) q->h1[j++] = q->h[q->h_len - i - 1];
    orc_windowcf_init(&q->w0, 2 * m);
    orc_windowcf_init(&q->w1, 2 * m);
    return q;
}

void orc_resamp2_crcf_reset(orc_resamp2_crcf *q)
{
    orc_windowcf_reset(&q->w0);
    orc_windowcf_reset(&q->w1);
}

void orc_resamp2_crcf_destroy(orc_resamp2_crcf *q)
{
    if (!q) return;
    orc_windowcf_free(&q->w0); orc_windowcf_free(&q->w1);
    free(q->h); free(q->h1); free(q);
}

void orc_resamp2_crcf_decim_execute(orc_resamp2_crcf *q, const cf32 *x, cf32 *y)
{
    /* filter branch on x[0] */
    orc_windowcf_push(&q->w1, x[0]);
    cf32 y1 = orc_dotprod_crcf(q->h1, orc_windowcf_read(&q->w1), q->h1_len);
    /* delay branch on x[1] */
    orc_windowcf_push(&q->w0, x[1]);
    cf32 y0 = orc_windowcf_read(&q->w0)[q->m - 1];
    *y = CMPLXF(crealf(y0) + crealf(y1), cimagf(y0) + cimagf(y1));
}

/* ------------------------------------------------------------------------------------------ */
/* msresamp2_crcf (decimator): liquid msresamp2.proto.c.  Stage design loop and the
 * "highest-index stage runs first" execution order per SURVEY A.3; output scaled by 1/2^stages. */

orc_msresamp2_crcf *orc_msresamp2_crcf_create_decim(unsigned num_stages, float fc, float f0, float As)
{
    orc_msresamp2_crcf *q = (orc_msresamp2_crcf *)calloc(1, sizeof(*q));
    q->num_stages = num_stages;
    q->M = 1u << num_stages;
    q->zeta = 1.0f / (float)q->M;
    q->buffer0 = (cf32 *)calloc(q->M, sizeof(cf32));
    q->buffer1 = (cf32 *)calloc(q->M, sizeof(cf32));
    q->fc_stage = (float *)calloc(num_stages + 1, sizeof(float));
    q->f0_stage = (float *)calloc(num_stages + 1, sizeof(float));
    q->As_stage = (float *)calloc(num_stages + 1, sizeof(float));
    q->m_stage = (unsigned *)calloc(num_stages + 1, sizeof(unsigned));
    q->stage = (orc_resamp2_crcf **)calloc(num_stages + 1, sizeof(*q->stage));
    float as = As + 5.0f;
    for (unsigned i = 0; i < num_stages; i++) {
        f0 = 0.5f * f0;
        fc = (i == 1) ? (0.5f - fc) / 2.0f : 0.5f * fc;
        float ft = 2.0f * (0.25f - fc);
        unsigned h_len = orc_estimate_req_filter_len(ft, as);
        unsigned m = (unsigned)ceilf((float)(h_len - 1) / 4.0f);
        q->fc_stage[i] = fc; q->f0_stage[i] = f0; q->As_stage[i] = as;
        q->m_stage[i] = m < 3 ? 3 : m;
    }
    for (unsigned i = 0; i < num_stages; i++)
        q->stage[i] = orc_resamp2_crcf_create(q->m_stage[i], q->f0_stage[i], q->As_stage[i]);
    return q;
}

void orc_msresamp2_crcf_reset(orc_msresamp2_crcf *q)
{
    for (unsigned i = 0; i < q->num_stages; i++) orc_resamp2_crcf_reset(q->stage[i]);
}

void orc_msresamp2_crcf_destroy(orc_msresamp2_crcf *q)
{
    if (!q) return;
    for (unsigned i = 0; i < q->num_stages; i++) orc_resamp2_crcf_destroy(q->stage[i]);
    free(q->stage); free(q->fc_stage); free(q->f0_stage); free(q->As_stage); free(q->m_stage);
    free(q->buffer0); free(q->buffer1); free(q);
}

void orc_msresamp2_crcf_decim_execute(orc_msresamp2_crcf *q, cf32 *x, cf32 *y)
{
    cf32 *b0 = x, *b1 = q->buffer1;
    for (unsigned s = 0; s < q->num_stages; s++) {
        unsigned g = q->num_stages - s - 1;
        unsigned k = 1u << g;
        for (unsigned i = 0; i < k; i++)
            orc_resamp2_crcf_decim_execute(q->stage[g], &b0[2 * i], &b1[i]);
        b0 = (s % 2) == 0 ? q->buffer1 : q->buffer0;
        b1 = (s % 2) == 0 ? q->buffer0 : q->buffer1;
    }
    *y = CMPLXF(crealf(b0[0]) * q->zeta, cimagf(b0[0]) * q->zeta);
}

/* ------------------------------------------------------------------------------------------ */
/* resamp_crcf, fixed-point phase variant: liquid resamp.fixed.proto.c + firpfb.proto.c.
 * bank of npfb sub-filters from firdes_kaiser(2*m*npfb+1, fc/npfb, As) normalised
 * so sum(h) = npfb; 24-bit phase per input sample; no interpolation between bank filters.       */

orc_resamp_crcf *orc_resamp_crcf_create(float rate, unsigned m, float fc, float As, unsigned npfb)
{
    orc_resamp_crcf *q = (orc_resamp_crcf *)calloc(1, sizeof(*q));
    unsigned bits = 0;
    while ((1u << bits) < npfb) bits++;
    q->rate = rate;
    q->step = (uint32_t)roundf((float)(1 << 24) / q->rate);
    q->m = m; q->fc = fc; q->As = As;
    q->bits_index = bits;
    q->npfb = 1u << bits;
    q->sub_len = 2 * m;
    unsigned n = 2 * q->m * q->npfb + 1;
    float *hf = (float *)calloc(n, sizeof(float));
    orc_firdes_kaiser(n, q->fc / (float)q->npfb, q->As, 0.0f, hf);
    float gain = 0.0f;
    for (unsigned i = 0; i < n; i++) gain += hf[i];
    gain = (float)q->npfb / gain;
    q->proto = (float *)calloc(n, sizeof(float));
    for (unsigned i = 0; i < n; i++) q->proto[i] = hf[i] * gain;
    free(hf);
    /* firpfb_create(npfb, h, n-1): sub-filter i = h[i + k*npfb], stored reversed */
    q->bank = (float *)calloc((size_t)q->npfb * q->sub_len, sizeof(float));
    for (unsigned i = 0; i < q->npfb; i++)
        for (unsigned k = 0; k < q->sub_len; k++)
            q->bank[(size_t)i * q->sub_len + (q->sub_len - k - 1)] = q->proto[i + k * q->npfb];
    orc_windowcf_init(&q->w, q->sub_len);
    q->phase = 0;
    return q;
}

void orc_resamp_crcf_reset(orc_resamp_crcf *q) { orc_windowcf_reset(&q->w); q->phase = 0; }

void orc_resamp_crcf_destroy(orc_resamp_crcf *q)
{
    if (!q) return;
    orc_windowcf_free(&q->w); free(q->bank); free(q->proto); free(q);
}

void orc_resamp_crcf_execute(orc_resamp_crcf *q, cf32 x, cf32 *y, unsigned *nw)
{
    orc_windowcf_push(&q->w, x);
    unsigned n = 0;
    while (q->phase <= 0x00ffffffu) {
        unsigned index = q->phase >> (24 - q->bits_index);
        y[n++] = orc_dotprod_crcf(q->bank + (size_t)index * q->sub_len, orc_windowcf_read(&q->w), q->sub_len);
        q->phase += q->step;
    }
    q->phase -= (1u << 24);
    *nw = n;
}

/* ------------------------------------------------------------------------------------------ */
/* msresamp_crcf: liquid msresamp.proto.c, decimation branch.  ref create src/sdr_pmr446.c:425-426
 * (rate = (float)SDR_RESAMPLERATE/SDR_SAMPLERATE, As = 60), run :796.  SURVEY A.3.               */

orc_msresamp_crcf *orc_msresamp_crcf_create(float rate, float As)
{
    if (!(rate > 0.0f) || rate > 1.0f) return NULL;   /* interpolation is not on this path */
    orc_msresamp_crcf *q = (orc_msresamp_crcf *)calloc(1, sizeof(*q));
    q->rate = rate; q->As = As;
    q->rate_arbitrary = rate; q->rate_halfband = 1.0f; q->num_halfband_stages = 0;
    while (q->rate_arbitrary < 0.5f) {
        q->num_halfband_stages++;
        q->rate_halfband *= 0.5f;
        q->rate_arbitrary *= 2.0f;
    }
    q->buffer = (cf32 *)calloc(4 + (1u << q->num_halfband_stages), sizeof(cf32));
    q->buffer_index = 0;
    q->halfband = orc_msresamp2_crcf_create_decim(q->num_halfband_stages, 0.4f, 0.0f, q->As);
    float fc = 0.515f * q->rate_arbitrary;
    if (fc > 0.49f) fc = 0.49f;
    q->arbitrary = orc_resamp_crcf_create(q->rate_arbitrary, 7, fc, q->As, 256);
    return q;
}

void orc_msresamp_crcf_reset(orc_msresamp_crcf *q)
{
    orc_msresamp2_crcf_reset(q->halfband);
    orc_resamp_crcf_reset(q->arbitrary);
    q->buffer_index = 0;
}

void orc_msresamp_crcf_destroy(orc_msresamp_crcf *q)
{
    if (!q) return;
    orc_msresamp2_crcf_destroy(q->halfband);
    orc_resamp_crcf_destroy(q->arbitrary);
    free(q->buffer); free(q);
}

void orc_msresamp_crcf_execute(orc_msresamp_crcf *q, const cf32 *x, unsigned nx, cf32 *y, unsigned *ny_out)
{
    unsigned M = 1u << q->num_halfband_stages, ny = 0, nw;
    cf32 hb;
    for (unsigned i = 0; i < nx; i++) {
        q->buffer[q->buffer_index++] = x[i];
        if (q->buffer_index == M) {
            orc_msresamp2_crcf_decim_execute(q->halfband, q->buffer, &hb);
            orc_resamp_crcf_execute(q->arbitrary, hb, &y[ny], &nw);
            ny += nw;
            q->buffer_index = 0;
        }
    }
    *ny_out = ny;
}

/* ------------------------------------------------------------------------------------------ */
/* msresamp_rrrf, interpolation branch: liquid msresamp.proto.c / msresamp2.proto.c / resamp2.proto.c /
 * resamp.fixed.proto.c with T = float.  ref create src/dsd_in.c:104 (rate = (float)48000/12500, As 60),
 * run :170.  Structure (restated from the library, same confidence as SURVEY A.3):
 *   create : while (rate_arbitrary > 2) { stages++; rate_arbitrary *= 0.5; }            3.84 -> 1 stage, 1.92
 *   execute: per input sample  resamp_execute -> nw outputs;  each through the half-band interpolators, stage 0
 *            (lowest rate) first, 2^stages outputs per arbitrary-resampler output; no output scaling (unity DC gain:
 *            the delay branch passes x, the filter branch taps sum to ~1).
 *   resamp2 interp_execute(x): push x into w0, y[0] = w0[m-1]; push x into w1, y[1] = dot(h1, w1).               */

orc_resamp2_rrrf *orc_resamp2_rrrf_create(unsigned m, float f0, float As)
{
    orc_resamp2_rrrf *q = (orc_resamp2_rrrf *)calloc(1, sizeof(*q));
    orc_resamp2_crcf *c = orc_resamp2_crcf_create(m, f0, As);      /* same prototype, same branch taps */
    q->m = m; q->h_len = c->h_len; q->h1_len = c->h1_len;
    q->h = (float *)calloc(q->h_len, sizeof(float));
    q->h1 = (float *)calloc(q->h1_len, sizeof(float));
    memcpy(q->h, c->h, q->h_len * sizeof(float));
    memcpy(q->h1, c->h1, q->h1_len * sizeof(float));
    orc_resamp2_crcf_destroy(c);
    orc_windowf_init(&q->w0, 2 * m);
    orc_windowf_init(&q->w1, 2 * m);
    return q;
}

void orc_resamp2_rrrf_reset(orc_resamp2_rrrf *q) { orc_windowf_reset(&q->w0); orc_windowf_reset(&q->w1); }

void orc_resamp2_rrrf_destroy(orc_resamp2_rrrf *q)
{
    if (!q) return;
    orc_windowf_free(&q->w0); orc_windowf_free(&q->w1);
    free(q->h); free(q->h1); free(q);
}

void orc_resamp2_rrrf_interp_execute(orc_resamp2_rrrf *q, float x, float *y)
{
    orc_windowf_push(&q->w0, x);                                    /* delay branch */
    y[0] = orc_windowf_read(&q->w0)[q->m - 1];
    orc_windowf_push(&q->w1, x);                                    /* filter branch */
    y[1] = orc_dotprod_rrrf(q->h1, orc_windowf_read(&q->w1), q->h1_len);
}

orc_resamp_rrrf *orc_resamp_rrrf_create(float rate, unsigned m, float fc, float As, unsigned npfb)
{
    orc_resamp_rrrf *q = (orc_resamp_rrrf *)calloc(1, sizeof(*q));
    orc_resamp_crcf *c = orc_resamp_crcf_create(rate, m, fc, As, npfb);   /* same bank design */
    q->m = c->m; q->npfb = c->npfb; q->bits_index = c->bits_index; q->sub_len = c->sub_len;
    q->rate = c->rate; q->fc = c->fc; q->As = c->As; q->step = c->step; q->phase = 0;
    const unsigned n = 2 * q->m * q->npfb + 1;
    q->proto = (float *)calloc(n, sizeof(float));
    q->bank = (float *)calloc((size_t)q->npfb * q->sub_len, sizeof(float));
    memcpy(q->proto, c->proto, n * sizeof(float));
    memcpy(q->bank, c->bank, (size_t)q->npfb * q->sub_len * sizeof(float));
    orc_resamp_crcf_destroy(c);
    orc_windowf_init(&q->w, q->sub_len);
    return q;
}

void orc_resamp_rrrf_reset(orc_resamp_rrrf *q) { orc_windowf_reset(&q->w); q->phase = 0; }

void orc_resamp_rrrf_destroy(orc_resamp_rrrf *q)
{
    if (!q) return;
    orc_windowf_free(&q->w); free(q->bank); free(q->proto); free(q);
}

void orc_resamp_rrrf_execute(orc_resamp_rrrf *q, float x, float *y, unsigned *nw)
{
    orc_windowf_push(&q->w, x);
    unsigned n = 0;
    while (q->phase <= 0x00ffffffu) {
        unsigned index = q->phase >> (24 - q->bits_index);
        y[n++] = orc_dotprod_rrrf(q->bank + (size_t)index * q->sub_len, orc_windowf_read(&q->w), q->sub_len);
        q->phase += q->step;
    }
    q->phase -= (1u << 24);
    *nw = n;
}

orc_msresamp_rrrf *orc_msresamp_rrrf_create(float rate, float As)
{
    if (!(rate >= 1.0f) || rate > 1024.0f) return NULL;   /* decimation of real streams is not on any reference path */
    orc_msresamp_rrrf *q = (orc_msresamp_rrrf *)calloc(1, sizeof(*q));
    q->rate = rate; q->As = As;
    q->rate_arbitrary = rate; q->rate_halfband = 1.0f; q->num_halfband_stages = 0;
    while (q->rate_arbitrary > 2.0f) {
        q->num_halfband_stages++;
        q->rate_halfband *= 2.0f;
        q->rate_arbitrary *= 0.5f;
    }
    const unsigned ns = q->num_halfband_stages;
    q->m_stage = (unsigned *)calloc(ns + 1, sizeof(unsigned));
    q->stage = (orc_resamp2_rrrf **)calloc(ns + 1, sizeof(*q->stage));
    {
        /* msresamp2_create(INTERP, ns, fc = 0.4, f0 = 0, As): same stage design loop as the decimator */
        orc_msresamp2_crcf *d = orc_msresamp2_crcf_create_decim(ns, 0.4f, 0.0f, As);
        for (unsigned i = 0; i < ns; i++) {
            q->m_stage[i] = d->m_stage[i];
            q->stage[i] = orc_resamp2_rrrf_create(d->m_stage[i], d->f0_stage[i], d->As_stage[i]);
        }
        orc_msresamp2_crcf_destroy(d);
    }
    float fc = 0.515f * q->rate_arbitrary;
    if (fc > 0.49f) fc = 0.49f;
    q->arbitrary = orc_resamp_rrrf_create(q->rate_arbitrary, 7, fc, q->As, 256);
    q->buffer0 = (float *)calloc((1u << ns) + 4, sizeof(float));
    q->buffer1 = (float *)calloc((1u << ns) + 4, sizeof(float));
    return q;
}

void orc_msresamp_rrrf_reset(orc_msresamp_rrrf *q)
{
    for (unsigned i = 0; i < q->num_halfband_stages; i++) orc_resamp2_rrrf_reset(q->stage[i]);
    orc_resamp_rrrf_reset(q->arbitrary);
}

void orc_msresamp_rrrf_destroy(orc_msresamp_rrrf *q)
{
    if (!q) return;
    for (unsigned i = 0; i < q->num_halfband_stages; i++) orc_resamp2_rrrf_destroy(q->stage[i]);
    orc_resamp_rrrf_destroy(q->arbitrary);
    free(q->stage); free(q->m_stage); free(q->buffer0); free(q->buffer1); free(q);
}

void orc_msresamp_rrrf_execute(orc_msresamp_rrrf *q, const float *x, unsigned nx, float *y, unsigned *ny_out)
{
    const unsigned ns = q->num_halfband_stages;
    unsigned ny = 0, nw;
    float arb[8];
    for (unsigned i = 0; i < nx; i++) {
        orc_resamp_rrrf_execute(q->arbitrary, x[i], arb, &nw);
        for (unsigned j = 0; j < nw; j++) {
            if (ns == 0) { y[ny++] = arb[j]; continue; }
            float *b0 = q->buffer0, *b1 = q->buffer1;
            b0[0] = arb[j];
            for (unsigned s = 0; s < ns; s++) {
                const unsigned k = 1u << s;
                float *dst = (s + 1 == ns) ? &y[ny] : b1;
                for (unsigned u = 0; u < k; u++) orc_resamp2_rrrf_interp_execute(q->stage[s], b0[u], &dst[2 * u]);
                float *t = b0; b0 = b1; b1 = t;
            }
            ny += 1u << ns;
        }
    }
    *ny_out = ny;
}

/* ------------------------------------------------------------------------------------------ */
/* nco_crcf (LIQUID_VCO): liquid nco.proto.c.  ref create/set src/sdr_pmr446.c:430-434, run
 * :810-811.  SURVEY A.4: 32-bit phase; VCO flavour evaluates sinf/cosf of the float phase.       */

uint32_t orc_nco_constrain(float theta)
{
    float p = theta * 0.159154943091895f;
    float fpart = p - (float)((long)p);
    if (fpart < 0.0f) fpart += 1.0f;
    return (uint32_t)(fpart * (float)0xffffffffu);
}

void orc_nco_set_frequency(orc_nco_crcf *q, float dtheta) { q->d_theta = orc_nco_constrain(dtheta); }
void orc_nco_reset(orc_nco_crcf *q) { q->theta = 0; }

void orc_nco_sincos(const orc_nco_crcf *q, float *s, float *c)
{
    float theta = (float)q->theta * (float)(2.0 * M_PI / 4294967296.0);
    *s = sinf(theta);
    *c = cosf(theta);
}

cf32 orc_nco_mix_down(const orc_nco_crcf *q, cf32 x)
{
    float s, c;
    orc_nco_sincos(q, &s, &c);
    /* y = x * conj(c + j s) */
    float xr = crealf(x), xi = cimagf(x);
    return CMPLXF(xr * c + xi * s, xi * c - xr * s);
}

/* ------------------------------------------------------------------------------------------ */
/* forward radix-2 DIT FFT, unscaled (liquid falls back to its internal FFT; any correct DFT
 * matches it to float rounding).  Twiddles evaluated in double, rounded to float.              */

orc_fft *orc_fft_create(unsigned n)
{
    unsigned l = 0;
    while ((1u << l) < n) l++;
    if ((1u << l) != n) return NULL;
    orc_fft *f = (orc_fft *)calloc(1, sizeof(*f));
    f->n = n; f->log2n = l;
    f->tw = (cf32 *)calloc(n / 2 + 1, sizeof(cf32));
    f->rev = (unsigned *)calloc(n, sizeof(unsigned));
    for (unsigned k = 0; k < n / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        f->tw[k] = CMPLXF((float)cos(a), (float)sin(a));
    }
    for (unsigned i = 0; i < n; i++) {
        unsigned r = 0;
        for (unsigned b = 0; b < l; b++) if (i & (1u << b)) r |= 1u << (l - 1 - b);
        f->rev[i] = r;
    }
    return f;
}

void orc_fft_destroy(orc_fft *f) { if (!f) return; free(f->tw); free(f->rev); free(f); }

void orc_fft_forward(const orc_fft *f, const cf32 *in, cf32 *out)
{
    unsigned n = f->n;
    for (unsigned i = 0; i < n; i++) out[f->rev[i]] = in[i];
    for (unsigned len = 2; len <= n; len <<= 1) {
        unsigned half = len >> 1, tstep = n / len;
        for (unsigned base = 0; base < n; base += len) {
            for (unsigned k = 0; k < half; k++) {
                cf32 w = f->tw[k * tstep];
                float wr = crealf(w), wi = cimagf(w);
                cf32 b = out[base + k + half], a = out[base + k];
                float br = crealf(b), bi = cimagf(b);
                float tr = br * wr - bi * wi, ti = br * wi + bi * wr;
                out[base + k]        = CMPLXF(crealf(a) + tr, cimagf(a) + ti);
                out[base + k + half] = CMPLXF(crealf(a) - tr, cimagf(a) - ti);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* firpfbch_crcf analyzer: liquid firpfbch.proto.c.  ref create src/sdr_pmr446.c:436-437
 * (LIQUID_ANALYZER, 16, 13, 80.0), run :814.  SURVEY A.5.                                        */

orc_firpfbch_crcf *orc_firpfbch_crcf_create_kaiser(unsigned M, unsigned m, float As)
{
    orc_fft *fft = orc_fft_create(M);
    if (!fft) return NULL;
    orc_firpfbch_crcf *q = (orc_firpfbch_crcf *)calloc(1, sizeof(*q));
    q->M = M; q->p = 2 * m; q->fft = fft;
    unsigned h_len = 2 * M * m + 1;
    q->h = (float *)calloc(h_len, sizeof(float));
    float fc = 0.5f / (float)M;
    orc_firdes_kaiser(h_len, fc, As, 0.0f, q->h);
    q->h_len = q->p * M;
    q->dp = (float *)calloc((size_t)M * q->p, sizeof(float));
    q->w = (orc_windowcf *)calloc(M, sizeof(orc_windowcf));
    for (unsigned i = 0; i < M; i++) {
        for (unsigned n = 0; n < q->p; n++) q->dp[(size_t)i * q->p + (q->p - n - 1)] = q->h[i + n * M];
        orc_windowcf_init(&q->w[i], q->p);
    }
    q->X = (cf32 *)calloc(M, sizeof(cf32));
    q->x = (cf32 *)calloc(M, sizeof(cf32));
    q->filter_index = M - 1;
    return q;
}

void orc_firpfbch_crcf_reset(orc_firpfbch_crcf *q)
{
    for (unsigned i = 0; i < q->M; i++) orc_windowcf_reset(&q->w[i]);
    q->filter_index = q->M - 1;
}

void orc_firpfbch_crcf_destroy(orc_firpfbch_crcf *q)
{
    if (!q) return;
    for (unsigned i = 0; i < q->M; i++) orc_windowcf_free(&q->w[i]);
    free(q->w); free(q->dp); free(q->h); free(q->X); free(q->x);
    orc_fft_destroy(q->fft); free(q);
}

void orc_firpfbch_crcf_analyzer_execute(orc_firpfbch_crcf *q, const cf32 *x, cf32 *y)
{
    for (unsigned i = 0; i < q->M; i++) {
        orc_windowcf_push(&q->w[q->filter_index], x[i]);
        q->filter_index = (q->filter_index + q->M - 1) % q->M;
    }
    for (unsigned i = 0; i < q->M; i++)
        q->X[q->M - i - 1] = orc_dotprod_crcf(q->dp + (size_t)i * q->p, orc_windowcf_read(&q->w[i]), q->p);
    orc_fft_forward(q->fft, q->X, q->x);
    memmove(y, q->x, q->M * sizeof(cf32));
}

/* ------------------------------------------------------------------------------------------ */
/* freqdem: liquid freqdem.proto.c.  ref create src/sdr_pmr446.c:440 (kf = 0.5), run :881,
 * reset :866.  SURVEY A.6: m = arg(conj(r') * r) / (2 pi kf).                                    */

void orc_freqdem_init(orc_freqdem *q, float kf)
{
    q->kf = kf;
    q->ref = 1.0f / (2.0f * (float)M_PI * kf);
    q->r_prime = 0;
}

void orc_freqdem_demodulate_block(orc_freqdem *q, const cf32 *r, unsigned n, float *m)
{
    for (unsigned k = 0; k < n; k++) {
        float pr = crealf(q->r_prime), pi = cimagf(q->r_prime);
        float cr = crealf(r[k]), ci = cimagf(r[k]);
        /* conj(p) * c = (pr*cr + pi*ci) + j (pr*ci - pi*cr) */
        float re = pr * cr + pi * ci;
        float im = pr * ci - pi * cr;
        m[k] = atan2f(im, re) * q->ref;
        q->r_prime = r[k];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* firfilt_rrrf: liquid firfilt.proto.c.  ref create src/sdr_pmr446.c:443-444 (377-tap HP),
 * :453-454 (103-tap LP), :458 (101-tap FIR de-emphasis); run :882, :896, :901.  SURVEY A.7.      */

orc_firfilt_rrrf *orc_firfilt_rrrf_create(const float *h, unsigned n)
{
    orc_firfilt_rrrf *q = (orc_firfilt_rrrf *)calloc(1, sizeof(*q));
    q->n = n;
    q->hr = (float *)calloc(n, sizeof(float));
    for (unsigned i = 0; i < n; i++) q->hr[i] = h[n - i - 1];
    q->scale = 1.0f;
    orc_windowf_init(&q->w, n);
    return q;
}

void orc_firfilt_rrrf_reset(orc_firfilt_rrrf *q) { orc_windowf_reset(&q->w); }

void orc_firfilt_rrrf_destroy(orc_firfilt_rrrf *q)
{
    if (!q) return;
    orc_windowf_free(&q->w); free(q->hr); free(q);
}

void orc_firfilt_rrrf_execute_block(orc_firfilt_rrrf *q, const float *x, unsigned n, float *y)
{
    for (unsigned k = 0; k < n; k++) {
        orc_windowf_push(&q->w, x[k]);
        y[k] = orc_dotprod_rrrf(q->hr, orc_windowf_read(&q->w), q->n) * q->scale;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* wdelayf: liquid wdelay.proto.c.  ref create src/sdr_pmr446.c:447 (delay 188), run :885-887.
 * push(x) then read() returns the sample pushed `delay` pushes earlier.              */

orc_wdelayf *orc_wdelayf_create(unsigned delay)
{
    orc_wdelayf *q = (orc_wdelayf *)calloc(1, sizeof(*q));
    q->delay = delay;
    q->v = (float *)calloc(delay + 1, sizeof(float));
    q->idx = 0;
    return q;
}

void orc_wdelayf_reset(orc_wdelayf *q) { memset(q->v, 0, (q->delay + 1) * sizeof(float)); q->idx = 0; }
void orc_wdelayf_destroy(orc_wdelayf *q) { if (!q) return; free(q->v); free(q); }

/* ------------------------------------------------------------------------------------------ */
/* cbuffercf: liquid cbuffer.proto.c.  ref create src/sdr_pmr446.c:467, use :797,804-805,815.
 * SURVEY A.9.                                                                                    */

orc_cbuffercf *orc_cbuffercf_create(unsigned max_size)
{
    orc_cbuffercf *q = (orc_cbuffercf *)calloc(1, sizeof(*q));
    q->max_size = max_size; q->max_read = max_size;
    q->num_alloc = q->max_size + q->max_read - 1;
    q->v = (cf32 *)calloc(q->num_alloc, sizeof(cf32));
    return q;
}

void orc_cbuffercf_reset(orc_cbuffercf *q) { q->num = 0; q->ri = 0; q->wi = 0; }
void orc_cbuffercf_destroy(orc_cbuffercf *q) { if (!q) return; free(q->v); free(q); }

int orc_cbuffercf_write(orc_cbuffercf *q, const cf32 *v, unsigned n)
{
    if (n > q->max_size - q->num) return 1;   /* liquid: LIQUID_EIRANGE, nothing written */
    q->num += n;
    unsigned k = q->max_size - q->wi;
    if (n > k) {
        memmove(q->v + q->wi, v, k * sizeof(cf32));
        memmove(q->v, v + k, (n - k) * sizeof(cf32));
        q->wi = n - k;
    } else {
        memmove(q->v + q->wi, v, n * sizeof(cf32));
        q->wi += n;
    }
    return 0;
}

void orc_cbuffercf_read(orc_cbuffercf *q, unsigned n, cf32 **v, unsigned *nread)
{
    if (n > q->num) n = q->num;
    if (n > q->max_size - q->ri)   /* linearise: copy the wrapped head behind the tail */
        memmove(q->v + q->max_size, q->v, (q->max_read - 1) * sizeof(cf32));
    *v = q->v + q->ri;
    *nread = n;
}

int orc_cbuffercf_release(orc_cbuffercf *q, unsigned n)
{
    if (n > q->num) return 1;
    q->ri = (q->ri + n) % q->max_size;
    q->num -= n;
    return 0;
}

/* ================= spgramcf / asgramcf (see orc_dsp.h) ================= */
orc_spgramcf *orc_spgramcf_create(unsigned nfft, unsigned window_len, unsigned delay)
{
    if (nfft < 2 || window_len == 0 || window_len > nfft || delay == 0) return NULL;
    orc_fft *fft = orc_fft_create(nfft);
    if (!fft) return NULL;
    orc_spgramcf *q = (orc_spgramcf *)calloc(1, sizeof(*q));
    q->nfft = nfft; q->window_len = window_len; q->delay = delay; q->fft = fft;
    q->w = (float *)calloc(window_len, sizeof(float));
    q->buf_time = (cf32 *)calloc(nfft, sizeof(cf32));
    q->buf_freq = (cf32 *)calloc(nfft, sizeof(cf32));
    q->psd = (float *)calloc(nfft, sizeof(float));
    orc_windowcf_init(&q->buffer, window_len);
    /* liquid_hann(i, wlen) = 0.5 - 0.5 cos(2 pi i / (wlen - 1)), then the scale by window magnitude and FFT size */
    float g = 0.0f;
    for (unsigned i = 0; i < window_len; i++) {
        q->w[i] = 0.5f - 0.5f * cosf((2.0f * (float)M_PI * (float)i) / ((float)(window_len - 1)));
        g += q->w[i] * q->w[i];
    }
    g = (float)M_SQRT2 / (sqrtf(g / (float)window_len) * sqrtf((float)nfft));
    for (unsigned i = 0; i < window_len; i++) q->w[i] *= g;
    orc_spgramcf_reset(q);
    return q;
}

void orc_spgramcf_destroy(orc_spgramcf *q)
{
    if (!q) return;
    orc_windowcf_free(&q->buffer); orc_fft_destroy(q->fft);
    free(q->w); free(q->buf_time); free(q->buf_freq); free(q->psd); free(q);
}

void orc_spgramcf_reset(orc_spgramcf *q)
{
    memset(q->buf_time, 0, q->nfft * sizeof(cf32));
    memset(q->psd, 0, q->nfft * sizeof(float));
    q->sample_timer = q->delay; q->num_transforms = 0;
    orc_windowcf_reset(&q->buffer);
}

static void spgram_step(orc_spgramcf *q)
{
    const cf32 *rc = orc_windowcf_read(&q->buffer);
    for (unsigned i = 0; i < q->window_len; i++) q->buf_time[i] = rc[i] * q->w[i];      /* the rest stays zero (padding) */
    orc_fft_forward(q->fft, q->buf_time, q->buf_freq);
    for (unsigned i = 0; i < q->nfft; i++) {
        const float re = crealf(q->buf_freq[i]), im = cimagf(q->buf_freq[i]);
        const float v = re * re + im * im;
        q->psd[i] = q->num_transforms == 0 ? v : q->psd[i] + v;                           /* alpha = 1: plain accumulation */
    }
    q->num_transforms++;
}

void orc_spgramcf_write(orc_spgramcf *q, const cf32 *x, unsigned n)
{
    for (unsigned i = 0; i < n; i++) {
        orc_windowcf_push(&q->buffer, x[i]);
        if (--q->sample_timer) continue;
        q->sample_timer = q->delay;
        spgram_step(q);
    }
}

void orc_spgramcf_get_psd(const orc_spgramcf *q, float *psd_db)
{
    const float scale = 1.0f / (float)(q->num_transforms ? q->num_transforms : 1);
    const unsigned h = q->nfft / 2;
    for (unsigned i = 0; i < q->nfft; i++) {
        float v = q->psd[(i + h) % q->nfft];
        if (v < 1e-12f) v = 1e-12f;
        psd_db[i] = 10.0f * log10f(v * scale);
    }
}

orc_asgramcf *orc_asgramcf_create(unsigned nfft)
{
    if (nfft < 2) return NULL;
    orc_asgramcf *q = (orc_asgramcf *)calloc(1, sizeof(*q));
    q->nfft = nfft; q->p = 4; q->nfftp = nfft * q->p;
    q->periodogram = orc_spgramcf_create(q->nfftp, nfft, nfft / 2);
    if (!q->periodogram) { free(q); return NULL; }
    q->psd = (float *)calloc(q->nfftp, sizeof(float));
    static const char lc[10] = {' ', '.', ',', '-', '+', '*', '&', 'N', 'M', '#'};
    q->num_levels = 10;
    memcpy(q->levelchar, lc, sizeof(lc));
    orc_asgramcf_set_scale(q, 0.0f, 10.0f);
    return q;
}

void orc_asgramcf_destroy(orc_asgramcf *q) { if (!q) return; orc_spgramcf_destroy(q->periodogram); free(q->psd); free(q); }

void orc_asgramcf_set_scale(orc_asgramcf *q, float ref, float div)
{
    q->ref = ref; q->div = div;
    for (unsigned i = 0; i < q->num_levels; i++) q->levels[i] = q->ref + (float)i * q->div;
}

void orc_asgramcf_write(orc_asgramcf *q, const cf32 *x, unsigned n) { orc_spgramcf_write(q->periodogram, x, n); }

void orc_asgramcf_execute(orc_asgramcf *q, char *ascii, float *peakval, float *peakfreq, float *psd_db_out)
{
    if (q->periodogram->num_transforms == 0) {
        memset(ascii, ' ', q->nfft);
        *peakval = 0.0f; *peakfreq = 0.0f;
        if (psd_db_out) for (unsigned i = 0; i < q->nfftp; i++) psd_db_out[i] = 0.0f;
        orc_spgramcf_reset(q->periodogram);
        return;
    }
    orc_spgramcf_get_psd(q->periodogram, q->psd);
    orc_spgramcf_reset(q->periodogram);
    if (psd_db_out) memcpy(psd_db_out, q->psd, q->nfftp * sizeof(float));
    for (unsigned i = 0; i < q->nfftp; i++)
        if (i == 0 || q->psd[i] > *peakval) { *peakval = q->psd[i]; *peakfreq = (float)i / (float)q->nfftp - 0.5f; }
    for (unsigned i = 0; i < q->nfft; i++) {
        float v = 0.0f;
        for (unsigned j = 0; j < q->p; j++) { const float x = q->psd[q->p * i + j]; v = (j == 0 || x > v) ? x : v; }
        ascii[i] = q->levelchar[0];
        for (unsigned j = 0; j < q->num_levels; j++) if (v > q->levels[j]) ascii[i] = q->levelchar[j];
    }
}
