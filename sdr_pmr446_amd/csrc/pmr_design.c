/* pmr_design.c -- coefficient design for the chain, the job of init_liquid() (reference
 * src/sdr_pmr446.c:420-480).  The liquid-dsp v1.7.0 constructors named there are restated from their
 * published algorithms: Kaiser-windowed sinc prototypes evaluated in double and
 * rounded once to float32.  Host-only C; the kernels receive the finished tables.
 */
#include "pmr_design.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Kaiser window pieces ---- */

static float kaiser_beta(float As)
{
    As = fabsf(As);
    if (As > 50.0f) return 0.1102f * (As - 8.7f);
    if (As > 21.0f) return 0.5842f * powf(As - 21.0f, 0.4f) + 0.07886f * (As - 21.0f);
    return 0.0f;
}

static double bessel_i0(double z)
{
    double sum = 1.0, term = 1.0, half = 0.5 * z;
    for (unsigned k = 1; k < 200; k++) {
        term *= half / (double)k;
        double sq = term * term;
        sum += sq;
        if (sq < 1e-20 * sum) break;
    }
    return sum;
}

static double kaiser_at(unsigned i, unsigned n, double beta, double i0_beta)
{
    double t = (double)i - (double)(n - 1) / 2.0;
    double r = 2.0 * t / (double)(n - 1);
    double a = 1.0 - r * r;
    if (a < 0.0) a = 0.0;
    return bessel_i0(beta * sqrt(a)) / i0_beta;
}

static double sinc_pi(double x)
{
    if (fabs(x) < 1e-12) return 1.0;
    return sin(M_PI * x) / (M_PI * x);
}

/* firdes_kaiser(n, fc, As, mu = 0) */
static void design_kaiser_lowpass(unsigned n, float fc, float As, float *h)
{
    double beta = (double)kaiser_beta(As);
    double i0b = bessel_i0(beta);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0 + 0.0;
        h[i] = (float)(sinc_pi(2.0 * (double)fc * t) * kaiser_at(i, n, beta, i0b));
    }
}

/* resamp2 prototype: sinc(t/2) * kaiser, length 4m+1 (f0 = 0 on this path) */
static void design_halfband(unsigned m, float As, float *h)
{
    unsigned n = 4 * m + 1;
    double beta = (double)kaiser_beta(As);
    double i0b = bessel_i0(beta);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0;
        h[i] = (float)(sinc_pi(t / 2.0) * kaiser_at(i, n, beta, i0b) * cos(2.0 * M_PI * t * 0.0));
    }
}

static uint32_t nco_constrain(float theta)
{
    /* liquid nco: fractional part of theta/(2 pi) mapped onto 32 bits */
    float p = theta * 0.159154943091895f;
    float frac = p - (float)((long)p);
    if (frac < 0.0f) frac += 1.0f;
    return (uint32_t)(frac * (float)0xffffffffu);
}

void pmr_design_free(pmr_design *d)
{
    for (unsigned g = 0; g < PMR_MAX_STAGES; g++) { free(d->hb_proto[g]); free(d->hb_h1[g]); }
    free(d->arb_proto); free(d->arb_bank); free(d->nco_cs);
    free(d->pfb_proto); free(d->pfb_taps_t); free(d->fft_tw);
    memset(d, 0, sizeof(*d));
}

int pmr_design_build(pmr_design *d, double fs_in, unsigned M, double channel_width_hz, float dc_alpha,
                     float resamp_As, unsigned pfb_m, float pfb_As, float fm_kf)
{
    memset(d, 0, sizeof(*d));
    if (M < 1 || (M & (M - 1)) != 0 || M > 4096 || pfb_m < 1 || pfb_m > 32 || !(fs_in > 0) ||
        !(channel_width_hz > 0) || !(fm_kf > 0))
        return 1;

    /* ---- msresamp_crcf_create(rate, As): :425-426, SURVEY A.3 ---- */
    float resamplerate = (float)((double)M * channel_width_hz);
    d->rate = resamplerate / (float)fs_in;
    if (!(d->rate > 0.0f) || d->rate > 1.0f) return 1;
    d->rate_arb = d->rate;
    d->num_stages = 0;
    while (d->rate_arb < 0.5f) { d->num_stages++; d->rate_arb *= 2.0f; }
    if (d->num_stages >= PMR_MAX_STAGES) return 1;
    d->decim = 1u << d->num_stages;
    d->zeta = 1.0f / (float)d->decim;

    /* msresamp2_crcf_create(DECIM, stages, fc = 0.4, f0 = 0, As): per-stage lengths */
    {
        float fc = 0.4f, as = resamp_As + 5.0f;
        for (unsigned g = 0; g < d->num_stages; g++) {
            fc = (g == 1) ? (0.5f - fc) / 2.0f : 0.5f * fc;
            float ft = 2.0f * (0.25f - fc);
            unsigned h_len = (unsigned)((as - 7.95f) / (14.26f * ft));
            unsigned m = (unsigned)ceilf((float)(h_len - 1) / 4.0f);
            if (m < 3) m = 3;
            d->m_stage[g] = m;
            d->hb_proto[g] = (float *)calloc(4 * m + 1, sizeof(float));
            d->hb_h1[g] = (float *)calloc(2 * m, sizeof(float));
            design_halfband(m, as, d->hb_proto[g]);
            /* branch filter = odd prototype taps, reversed: h1[j] = h[4m-1-2j]; applied oldest-first */
            for (unsigned j = 0; j < 2 * m; j++) d->hb_h1[g][j] = d->hb_proto[g][4 * m - 1 - 2 * j];
        }
    }

    /* resamp_crcf_create(rate_arb, 7, min(0.515 r, 0.49), As, 256) */
    {
        float fc = 0.515f * d->rate_arb;
        if (fc > 0.49f) fc = 0.49f;
        d->arb_step = (uint32_t)roundf((float)(1 << 24) / d->rate_arb);
        unsigned n = 2 * PMR_ARB_M * PMR_ARB_NPFB + 1;
        float *hf = (float *)calloc(n, sizeof(float));
        design_kaiser_lowpass(n, fc / (float)PMR_ARB_NPFB, resamp_As, hf);
        float gain = 0.0f;
        for (unsigned i = 0; i < n; i++) gain += hf[i];
        gain = (float)PMR_ARB_NPFB / gain;
        d->arb_proto = (float *)calloc(n, sizeof(float));
        for (unsigned i = 0; i < n; i++) d->arb_proto[i] = hf[i] * gain;
        free(hf);
        unsigned L = 2 * PMR_ARB_M;
        d->arb_bank = (float *)calloc((size_t)PMR_ARB_NPFB * L, sizeof(float));
        for (unsigned i = 0; i < PMR_ARB_NPFB; i++)
            for (unsigned k = 0; k < L; k++)      /* k = 0 multiplies the oldest of the 2m samples */
                d->arb_bank[(size_t)i * L + k] = d->arb_proto[i + (L - 1 - k) * PMR_ARB_NPFB];
    }

    /* ---- nco_crcf (LIQUID_VCO), frequency of :432-434 ---- */
    {
        float offset = -0.5f * (float)(M - 1) / (float)M * 2 * M_PI;
        d->nco_dtheta = nco_constrain(offset);
        uint64_t g = 1;                                /* gcd(dtheta, 2^32) = lowest set bit */
        if (d->nco_dtheta == 0) g = 4294967296ull;
        else while (!(d->nco_dtheta & g)) g <<= 1;
        uint64_t period = 4294967296ull / g;
        if (period <= PMR_NCO_MAX_PERIOD) {
            d->nco_period = (unsigned)period;
            d->nco_cs = (float *)calloc((size_t)period * 2, sizeof(float));
            uint32_t th = 0;
            for (unsigned k = 0; k < d->nco_period; k++) {
                float thf = (float)th * (float)(2.0 * M_PI / 4294967296.0);
                d->nco_cs[2 * k] = cosf(thf);
                d->nco_cs[2 * k + 1] = sinf(thf);
                th += d->nco_dtheta;
            }
        } else {
            return 1;   /* non-power-of-two style offsets would need in-kernel sincos; not on this path */
        }
    }

    /* ---- firpfbch_crcf_create_kaiser(ANALYZER, M, m, As): :436-437, SURVEY A.5 ---- */
    {
        d->M = M; d->pfb_m = pfb_m; d->pfb_p = 2 * pfb_m;
        unsigned n = 2 * M * pfb_m + 1, p = d->pfb_p;
        d->pfb_proto = (float *)calloc(n, sizeof(float));
        design_kaiser_lowpass(n, 0.5f / (float)M, pfb_As, d->pfb_proto);
        d->pfb_taps_t = (float *)calloc((size_t)p * M, sizeof(float));
        for (unsigned k = 0; k < p; k++)
            for (unsigned c = 0; c < M; c++)
                d->pfb_taps_t[(size_t)k * M + c] = d->pfb_proto[(M - 1 - c) + (p - 1 - k) * M];
        d->fft_tw = (float *)calloc((size_t)M, sizeof(float));   /* M/2 complex */
        for (unsigned k = 0; k < M / 2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)M;
            d->fft_tw[2 * k] = (float)cos(a);
            d->fft_tw[2 * k + 1] = (float)sin(a);
        }
    }

    /* ---- iirfilt_crcf_create_dc_blocker(alpha): :422, SURVEY A.2 ---- */
    d->dc_a1 = -1.0f + dc_alpha;
    d->dc_lambda = -(double)d->dc_a1;

    /* ---- freqdem_create(kf): :440 ---- */
    d->fm_ref = 1.0f / (2.0f * (float)M_PI * fm_kf);

    /* ---- de-emphasis IIR literals of :462-463 (= scripts/filter_des.py:31-44, tau 50 us, fs 12.5 kHz) ---- */
    {
        float b[2] = {0.507301437230636, 0.507301437230636};
        float a[2] = {1.0, 0.014602874461272194};
        d->de_b0 = b[0] / a[0]; d->de_b1 = b[1] / a[0]; d->de_a1 = a[1] / a[0];
    }
    return 0;
}

/* msresamp_rrrf_create(rate, As), interpolation (reference src/dsd_in.c:104): the same constructors as above */
int pmr_up_design_build(pmr_up_design *u, float rate, float As)
{
    memset(u, 0, sizeof(*u));
    if (!(rate >= 1.0f) || rate > 256.0f) return 1;
    u->rate = rate; u->rate_arb = rate; u->num_stages = 0;
    while (u->rate_arb > 2.0f) { u->num_stages++; u->rate_arb *= 0.5f; }
    if (u->num_stages > PMR_UP_MAX_STAGES) return 1;
    {
        float fc = 0.4f, as = As + 5.0f;
        for (unsigned g = 0; g < u->num_stages; g++) {
            fc = (g == 1) ? (0.5f - fc) / 2.0f : 0.5f * fc;
            float ft = 2.0f * (0.25f - fc);
            unsigned h_len = (unsigned)((as - 7.95f) / (14.26f * ft));
            unsigned m = (unsigned)ceilf((float)(h_len - 1) / 4.0f);
            if (m < 3) m = 3;
            u->m_stage[g] = m;
            float *proto = (float *)calloc(4 * m + 1, sizeof(float));
            u->hb_h1[g] = (float *)calloc(2 * m, sizeof(float));
            design_halfband(m, as, proto);
            for (unsigned j = 0; j < 2 * m; j++) u->hb_h1[g][j] = proto[4 * m - 1 - 2 * j];
            free(proto);
        }
    }
    {
        float fc = 0.515f * u->rate_arb;
        if (fc > 0.49f) fc = 0.49f;
        u->arb_step = (uint32_t)roundf((float)(1 << 24) / u->rate_arb);
        unsigned n = 2 * PMR_ARB_M * PMR_ARB_NPFB + 1, L = 2 * PMR_ARB_M;
        float *hf = (float *)calloc(n, sizeof(float));
        design_kaiser_lowpass(n, fc / (float)PMR_ARB_NPFB, As, hf);
        float gain = 0.0f;
        for (unsigned i = 0; i < n; i++) gain += hf[i];
        gain = (float)PMR_ARB_NPFB / gain;
        u->arb_bank = (float *)calloc((size_t)PMR_ARB_NPFB * L, sizeof(float));
        for (unsigned i = 0; i < PMR_ARB_NPFB; i++)
            for (unsigned k = 0; k < L; k++)
                u->arb_bank[(size_t)i * L + k] = (hf[i + (L - 1 - k) * PMR_ARB_NPFB] * gain);
        free(hf);
    }
    return 0;
}

void pmr_up_design_free(pmr_up_design *u)
{
    for (unsigned g = 0; g < PMR_UP_MAX_STAGES; g++) free(u->hb_h1[g]);
    free(u->arb_bank);
    memset(u, 0, sizeof(*u));
}

void pmr_design_buffer_sizes(const pmr_design *d, unsigned max_block, unsigned *res_size, unsigned *chan_size)
{
    /* :730-732: ceilf(1 + 2*chunk*rate), then integer division by M inside ceilf */
    unsigned rs = (unsigned)ceilf(1 + 2 * (float)max_block * d->rate);
    unsigned cs = (unsigned)ceilf((float)(rs / d->M));
    if (cs < 1) cs = 1;
    *res_size = rs; *chan_size = cs;
}
