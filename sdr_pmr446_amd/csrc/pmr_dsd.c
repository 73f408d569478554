/* pmr_dsd.c -- host side of the `dsd_in` chain (include/pmr_dsd.h; reference src/dsd_in.c:95-178).
 *
 * The dc-block + msresamp_crcf front end is the one the channelizer app uses (a pmr_chain in front-end-only mode, see
 * pmr_internal.h): at the reference's 1.024 MS/s -> 12.5 kS/s that is a six-stage half-band cascade, i.e. the two-level
 * fused front end.  What follows it runs at <= 48 kS/s: discriminator, arbitrary resampler, half-band interpolators
 * (pmr_dsd_kernels.hip).  All sample counts are closed-form in the number of raw samples consumed, so nothing is read back from
 * the device to size a launch.  No CPU fallback: without a HIP device pmr_dsd_create() fails.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pmr_dsd.h"
#include "pmr_design.h"
#include "pmr_internal.h"
#include "pmr_kernels.h"

struct pmr_dsd_s {
    pmr_dsd_cfg cfg;
    pmr_chain fe;                    /* dc-block + msresamp_crcf (:167-168)                              */
    pmr_fe_view v;
    pmr_up_design up;                /* msresamp_rrrf (:104)                                             */
    float fm_ref;
    unsigned res_size, out_size;     /* :140-141                                                          */
    float *d_fm; uint64_t fm_mask;   /* discriminator ring, absolute resampled-sample index               */
    float *d_u[PMR_UP_MAX_STAGES + 1]; uint64_t u_mask[PMR_UP_MAX_STAGES + 1];   /* u_0 = arbitrary resampler output */
    float *d_bank, *d_h1[PMR_UP_MAX_STAGES];
    int16_t *d_pcm; float *d_audio;
    uint64_t n_res;                  /* resampled samples produced since reset                            */
    uint64_t n_u;                    /* arbitrary-resampler outputs produced since reset                  */
    unsigned last_ny;
    char err[256];
};

static int dfail(pmr_dsd q, int code, const char *what, hipError_t e)
{
    if (q) snprintf(q->err, sizeof(q->err), "%s%s%s", what, e != hipSuccess ? ": " : "",
                    e != hipSuccess ? hipGetErrorString(e) : "");
    return code;
}
#define DCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return dfail(q, PMR_EHIP, what, e_); } while (0)

void pmr_dsd_default_cfg(pmr_dsd_cfg *c)
{
    memset(c, 0, sizeof(*c));
    c->fs_in = 1024000.0;            /* include/dsd_in.h:11 */
    c->sig_rate = 12500.0;           /* src/dsd_in.c:23 */
    c->audio_rate = 48000.0;         /* :22 */
    c->dcblock_alpha = 0.0005f;      /* :97 */
    c->resamp_As = 60.0f;            /* :100, :104 */
    c->fm_kf = 0.5f;                 /* :108 */
    c->max_block = 200000;           /* :25 */
    c->device = -1;
}

static int cfg_ok(const pmr_dsd_cfg *c)
{
    return c && c->fs_in > 0 && c->sig_rate > 0 && c->sig_rate <= c->fs_in && c->audio_rate >= c->sig_rate &&
           c->max_block > 0 && c->fm_kf > 0;
}

/* the front end of this chain expressed as a one-channel pmr_chain configuration */
static void fe_cfg(const pmr_dsd_cfg *c, pmr_chain_cfg *f)
{
    pmr_chain_default_cfg(f);
    f->fs_in = c->fs_in; f->num_channels = 1; f->channel_width_hz = c->sig_rate;
    f->dcblock_alpha = c->dcblock_alpha; f->resamp_As = c->resamp_As; f->fm_kf = c->fm_kf;
    f->max_block = c->max_block; f->device = c->device;
}

static float up_rate(const pmr_dsd_cfg *c) { return (float)c->audio_rate / (float)c->sig_rate; }   /* :104 */

static void sizes(const pmr_dsd_cfg *c, unsigned *res_size, unsigned *out_size)
{
    const float r_down = (float)c->sig_rate / (float)c->fs_in;
    *res_size = (unsigned)ceilf(1 + 2 * (float)c->max_block * r_down);       /* :140 */
    *out_size = (unsigned)ceilf(1 + 2 * (float)*res_size * up_rate(c));      /* :141 */
}

/* arbitrary-resampler outputs after A input samples: #{j : j*step < A*2^24} */
static uint64_t up_count(uint64_t A, uint32_t step) { return ((A << 24) + step - 1) / step; }

static int ring_alloc(pmr_dsd q, float **p, uint64_t *mask, uint64_t need)
{
    uint64_t cap = 1;
    while (cap < need) cap <<= 1;
    *mask = cap - 1;
    hipError_t e = hipMalloc((void **)p, (size_t)cap * sizeof(float));
    if (e != hipSuccess) return dfail(q, PMR_ENOMEM, "hipMalloc", e);
    e = hipMemset(*p, 0, (size_t)cap * sizeof(float));
    return e == hipSuccess ? PMR_OK : dfail(q, PMR_EHIP, "hipMemset", e);
}

static int upload(pmr_dsd q, float **p, const float *src, size_t n)
{
    hipError_t e = hipMalloc((void **)p, n * sizeof(float));
    if (e != hipSuccess) return dfail(q, PMR_ENOMEM, "hipMalloc", e);
    e = hipMemcpy(*p, src, n * sizeof(float), hipMemcpyHostToDevice);
    return e == hipSuccess ? PMR_OK : dfail(q, PMR_EHIP, "hipMemcpy", e);
}

pmr_dsd pmr_dsd_create(const pmr_dsd_cfg *cfg)
{
    if (!cfg_ok(cfg)) { fprintf(stderr, "pmr_dsd_create: invalid configuration\n"); return NULL; }
    pmr_dsd q = (pmr_dsd)calloc(1, sizeof(*q));
    if (!q) return NULL;
    q->cfg = *cfg;
    if (pmr_up_design_build(&q->up, up_rate(cfg), cfg->resamp_As)) {
        fprintf(stderr, "pmr_dsd_create: invalid interpolation rate\n");
        free(q); return NULL;
    }
    pmr_chain_cfg f;
    fe_cfg(cfg, &f);
    q->fe = pmr_chain_create_frontend(&f);          /* fails without a HIP device */
    if (!q->fe) { pmr_up_design_free(&q->up); free(q); return NULL; }
    pmr_chain_frontend_view(q->fe, &q->v);
    q->fm_ref = 1.0f / (2.0f * (float)M_PI * cfg->fm_kf);
    sizes(cfg, &q->res_size, &q->out_size);
    int rc = ring_alloc(q, &q->d_fm, &q->fm_mask, (uint64_t)q->res_size + 64);
    const unsigned S = q->up.num_stages;
    uint64_t n_stage = (uint64_t)((double)q->res_size * (double)q->up.rate_arb) + 8;    /* per-block outputs of u_0 */
    for (unsigned g = 0; g <= S && !rc; g++) {
        if (g < S) rc = ring_alloc(q, &q->d_u[g], &q->u_mask[g], n_stage + 2 * q->up.m_stage[g] + 64);
        n_stage *= 2;
    }
    if (!rc) rc = upload(q, &q->d_bank, q->up.arb_bank, (size_t)PMR_ARB_NPFB * 2 * PMR_ARB_M);
    for (unsigned g = 0; g < S && !rc; g++) rc = upload(q, &q->d_h1[g], q->up.hb_h1[g], 2 * q->up.m_stage[g]);
    if (!rc && hipMalloc((void **)&q->d_pcm, (size_t)q->out_size * sizeof(int16_t)) != hipSuccess) rc = PMR_ENOMEM;
    if (!rc && hipMalloc((void **)&q->d_audio, (size_t)q->out_size * sizeof(float)) != hipSuccess) rc = PMR_ENOMEM;
    if (rc) {
        fprintf(stderr, "pmr_dsd_create: %s\n", q->err);
        pmr_dsd_destroy(q);
        return NULL;
    }
    return q;
}

int pmr_dsd_destroy(pmr_dsd q)
{
    if (!q) return PMR_OK;
    if (q->fe) { pmr_chain_synchronize(q->fe); hipSetDevice(q->v.device); }
    void *bufs[] = { q->d_fm, q->d_bank, q->d_pcm, q->d_audio };
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); i++) if (bufs[i]) hipFree(bufs[i]);
    for (unsigned g = 0; g <= PMR_UP_MAX_STAGES; g++) if (q->d_u[g]) hipFree(q->d_u[g]);
    for (unsigned g = 0; g < PMR_UP_MAX_STAGES; g++) if (q->d_h1[g]) hipFree(q->d_h1[g]);
    if (q->fe) pmr_chain_destroy(q->fe);
    pmr_up_design_free(&q->up);
    free(q);
    return PMR_OK;
}

int pmr_dsd_synchronize(pmr_dsd q)
{
    if (!q) return PMR_EINVAL;
    int rc = pmr_chain_synchronize(q->fe);
    return rc ? dfail(q, rc, pmr_chain_last_error(q->fe), hipSuccess) : PMR_OK;
}

int pmr_dsd_reset(pmr_dsd q)
{
    if (!q) return PMR_EINVAL;
    int rc = pmr_chain_reset(q->fe);
    if (rc) return dfail(q, rc, pmr_chain_last_error(q->fe), hipSuccess);
    DCHK(hipMemset(q->d_fm, 0, (size_t)(q->fm_mask + 1) * sizeof(float)), "reset");
    for (unsigned g = 0; g < q->up.num_stages; g++)
        DCHK(hipMemset(q->d_u[g], 0, (size_t)(q->u_mask[g] + 1) * sizeof(float)), "reset");
    q->n_res = 0; q->n_u = 0; q->last_ny = 0;
    return PMR_OK;
}

unsigned pmr_dsd_max_out(pmr_dsd q) { return q ? q->out_size : 0; }
const char *pmr_dsd_last_error(pmr_dsd q) { return q ? q->err : "null handle"; }

int pmr_dsd_process_block_device(pmr_dsd q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio, unsigned cap,
                                 unsigned *n_out)
{
    if (!q) return PMR_EINVAL;
    if (n_in > q->cfg.max_block) return dfail(q, PMR_ERANGE, "n_in > max_block", hipSuccess);
    DCHK(hipSetDevice(q->v.device), "hipSetDevice");
    /* every capacity check runs on the closed-form plan BEFORE any state is advanced: a refused block is not consumed */
    const unsigned S = q->up.num_stages;
    const unsigned ny_plan = pmr_chain_plan_resampled(q->fe, n_in);
    const uint64_t j0 = q->n_u, j1 = up_count(q->n_res + ny_plan, q->up.arb_step);
    const unsigned nu = (unsigned)(j1 - j0), nz = nu << S;
    if (n_out) *n_out = nz;
    if (nz > q->out_size) return dfail(q, PMR_ERANGE, "output overflow", hipSuccess);
    if (nz > cap && (d_pcm || d_audio)) return dfail(q, PMR_ERANGE, "cap < samples produced", hipSuccess);
    unsigned ny = 0; uint64_t a0 = 0;
    int rc = pmr_chain_frontend_block(q->fe, d_iq, n_in, &ny, &a0);           /* :167-168 */
    if (rc) return dfail(q, rc, pmr_chain_last_error(q->fe), hipSuccess);
    if (a0 != q->n_res || ny != ny_plan) return dfail(q, PMR_EINVAL, "internal: resampled count mismatch", hipSuccess);
    q->n_res += ny; q->n_u = j1; q->last_ny = ny;
    pmr_stream_t st = (pmr_stream_t)q->v.stream_fe;
    if ((rc = pmr_launch_dsd_fm(st, q->v.d_xr, q->v.xr_mask, a0, ny, q->d_fm, q->fm_mask, q->fm_ref)))        /* :169 */
        return dfail(q, PMR_EHIP, "k_dsd_fm", (hipError_t)rc);
    if ((rc = pmr_launch_dsd_arb(st, q->d_fm, q->fm_mask, j0, nu, q->up.arb_step, q->d_bank, S ? q->d_u[0] : NULL,
                                 q->u_mask[0], S ? NULL : (int16_t *)d_pcm, S ? NULL : (float *)d_audio)))    /* :170 */
        return dfail(q, PMR_EHIP, "k_dsd_arb", (hipError_t)rc);
    uint64_t i0 = j0; unsigned n = nu;
    for (unsigned g = 0; g < S; g++) {
        const int last = g + 1 == S;
        if ((rc = pmr_launch_dsd_hb(st, q->d_u[g], q->u_mask[g], i0, n, (int)q->up.m_stage[g], q->d_h1[g],
                                    last ? NULL : q->d_u[g + 1], q->u_mask[g + 1], last ? (int16_t *)d_pcm : NULL,
                                    last ? (float *)d_audio : NULL)))
            return dfail(q, PMR_EHIP, "k_dsd_hb", (hipError_t)rc);
        i0 *= 2; n *= 2;
    }
    return PMR_OK;
}

int pmr_dsd_process_block(pmr_dsd q, const pmr_cf32 *iq, unsigned n_in, int16_t *pcm, float *audio, unsigned cap,
                          unsigned *n_out)
{
    if (!q) return PMR_EINVAL;
    if (n_in > q->cfg.max_block) return dfail(q, PMR_ERANGE, "n_in > max_block", hipSuccess);
    if (n_in && !iq) return dfail(q, PMR_EINVAL, "null input", hipSuccess);
    DCHK(hipSetDevice(q->v.device), "hipSetDevice");
    hipStream_t st = (hipStream_t)q->v.stream_fe;
    if (n_in) DCHK(hipMemcpyAsync(q->v.d_in, iq, (size_t)n_in * 8, hipMemcpyHostToDevice, st), "H2D");
    unsigned nz = 0;
    {   /* the caller's capacity is checked before the block is consumed */
        const uint64_t j1 = up_count(q->n_res + pmr_chain_plan_resampled(q->fe, n_in), q->up.arb_step);
        nz = (unsigned)(j1 - q->n_u) << q->up.num_stages;
        if (n_out) *n_out = nz;
        if (nz > cap && (pcm || audio)) return dfail(q, PMR_ERANGE, "cap < samples produced", hipSuccess);
    }
    int rc = pmr_dsd_process_block_device(q, q->v.d_in, n_in, pcm ? q->d_pcm : NULL, audio ? q->d_audio : NULL,
                                          q->out_size, &nz);
    if (rc) return rc;
    if (n_out) *n_out = nz;
    if (nz && pcm) DCHK(hipMemcpyAsync(pcm, q->d_pcm, (size_t)nz * sizeof(int16_t), hipMemcpyDeviceToHost, st), "D2H pcm");
    if (nz && audio) DCHK(hipMemcpyAsync(audio, q->d_audio, (size_t)nz * sizeof(float), hipMemcpyDeviceToHost, st), "D2H audio");
    return pmr_dsd_synchronize(q);
}

int pmr_dsd_debug_read(pmr_dsd q, int what, void *host_buf, size_t cap_bytes, size_t *n_bytes)
{
    if (!q || (what != 0 && what != 1)) return PMR_EINVAL;
    int rc = pmr_dsd_synchronize(q);
    if (rc) return rc;
    const size_t elem = what == 0 ? 8 : 4, n = q->last_ny;
    if (n_bytes) *n_bytes = n * elem;
    if (!host_buf || !n) return PMR_OK;
    if (cap_bytes < n * elem) return dfail(q, PMR_ERANGE, "debug buffer too small", hipSuccess);
    const char *ring = what == 0 ? (const char *)q->v.d_xr : (const char *)q->d_fm;
    const uint64_t mask = what == 0 ? q->v.xr_mask : q->fm_mask, capn = mask + 1, i0 = (q->n_res - n) & mask;
    const size_t first = (size_t)((capn - i0) < n ? (capn - i0) : n);
    DCHK(hipMemcpy(host_buf, ring + i0 * elem, first * elem, hipMemcpyDeviceToHost), "debug D2H");
    if (first < n) DCHK(hipMemcpy((char *)host_buf + first * elem, ring, (n - first) * elem, hipMemcpyDeviceToHost), "debug D2H");
    return PMR_OK;
}

/* ---- host-only helpers ---- */

int pmr_dsd_plan_block(const pmr_dsd_cfg *cfg, pmr_dsd_plan_state *st, unsigned n_in, unsigned *n_resampled,
                       unsigned *n_out)
{
    if (!cfg_ok(cfg) || !st) return PMR_EINVAL;
    pmr_chain_cfg f;
    fe_cfg(cfg, &f);
    pmr_plan_state ps = { st->n_raw, st->down_phase, 0 };
    unsigned ny = 0, ns = 0;
    int rc = pmr_cfg_plan_block(&f, &ps, n_in, &ny, &ns);
    if (rc) return rc;
    pmr_up_design u;
    if (pmr_up_design_build(&u, up_rate(cfg), cfg->resamp_As)) return PMR_EINVAL;
    const uint64_t j0 = up_count(st->n_resampled, u.arb_step), j1 = up_count(st->n_resampled + ny, u.arb_step);
    if (n_resampled) *n_resampled = ny;
    if (n_out) *n_out = (unsigned)(j1 - j0) << u.num_stages;
    st->n_raw = ps.n_raw; st->down_phase = ps.arb_phase; st->n_resampled += ny;
    pmr_up_design_free(&u);
    return PMR_OK;
}

unsigned pmr_dsd_cfg_info(const pmr_dsd_cfg *cfg, int what)
{
    if (!cfg_ok(cfg)) return 0;
    pmr_chain_cfg f;
    fe_cfg(cfg, &f);
    if (what == 0) return pmr_cfg_info(&f, PMR_INFO_NUM_STAGES, 0);
    if (what == 3) return pmr_cfg_info(&f, PMR_INFO_ARB_STEP, 0);
    pmr_up_design u;
    if (pmr_up_design_build(&u, up_rate(cfg), cfg->resamp_As)) return 0;
    unsigned r = 0;
    if (what == 1) r = u.num_stages;
    else if (what == 2) r = u.arb_step;
    else if (what >= 4 && (unsigned)(what - 4) < u.num_stages) r = u.m_stage[what - 4];
    pmr_up_design_free(&u);
    return r;
}
