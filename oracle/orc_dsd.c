/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md and orc_dsd.h). */
#include "orc_dsd.h"
#include "orc_chain.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_dsd_default_cfg(orc_dsd_cfg *c)
{
    memset(c, 0, sizeof(*c));
    c->fs_in = 1024000.0;       /* include/dsd_in.h:11 */
    c->sig_rate = 12500.0;      /* src/dsd_in.c:23 */
    c->audio_rate = 48000.0;    /* :22 */
    c->dcblock_alpha = 0.0005f; /* :97 */
    c->resamp_As = 60.0f;       /* :100,:104 */
    c->fm_kf = 0.5f;            /* :108 */
    c->max_block = 200000;      /* :25 */
}

orc_dsd *orc_dsd_create(const orc_dsd_cfg *cfg)
{
    if (!cfg || !(cfg->fs_in > 0) || !(cfg->sig_rate > 0) || !(cfg->audio_rate >= cfg->sig_rate) ||
        cfg->sig_rate > cfg->fs_in || cfg->max_block == 0)
        return NULL;
    orc_dsd *q = (orc_dsd *)calloc(1, sizeof(*q));
    q->cfg = *cfg;
    const float r_down = (float)cfg->sig_rate / (float)cfg->fs_in;        /* :100 ((float)SIG)/SDR */
    const float r_up = (float)cfg->audio_rate / (float)cfg->sig_rate;     /* :104 */
    q->res_size = (unsigned)ceilf(1 + 2 * (float)cfg->max_block * r_down);   /* :140 */
    q->out_size = (unsigned)ceilf(1 + 2 * (float)q->res_size * r_up);         /* :141 */
    q->dcblock = orc_iirfilt_crcf_create_dc_blocker(cfg->dcblock_alpha);      /* :97 */
    q->res_down = orc_msresamp_crcf_create(r_down, cfg->resamp_As);           /* :100 */
    q->res_up = orc_msresamp_rrrf_create(r_up, cfg->resamp_As);               /* :104 */
    orc_freqdem_init(&q->fm_demod, cfg->fm_kf);                               /* :108 */
    if (!q->dcblock || !q->res_down || !q->res_up) { orc_dsd_destroy(q); return NULL; }
    q->buffp = (cf32 *)calloc(cfg->max_block, sizeof(cf32));
    q->resamp_buf = (cf32 *)calloc(q->res_size, sizeof(cf32));
    q->fm_out_buf = (float *)calloc(q->res_size, sizeof(float));
    q->out_buf = (float *)calloc(q->out_size, sizeof(float));
    return q;
}

int orc_dsd_reset(orc_dsd *q)
{
    if (!q) return 1;
    orc_iirfilt_crcf_reset(q->dcblock);
    orc_msresamp_crcf_reset(q->res_down);
    orc_msresamp_rrrf_reset(q->res_up);
    orc_freqdem_reset(&q->fm_demod);
    return 0;
}

int orc_dsd_destroy(orc_dsd *q)
{
    if (!q) return 0;
    orc_iirfilt_crcf_destroy(q->dcblock);
    orc_msresamp_crcf_destroy(q->res_down);
    orc_msresamp_rrrf_destroy(q->res_up);
    free(q->buffp); free(q->resamp_buf); free(q->fm_out_buf); free(q->out_buf);
    free(q);
    return 0;
}

unsigned orc_dsd_max_out(const orc_dsd *q) { return q ? q->out_size : 0; }
unsigned orc_dsd_max_resampled(const orc_dsd *q) { return q ? q->res_size : 0; }

unsigned orc_dsd_info(const orc_dsd *q, int what)
{
    if (!q) return 0;
    switch (what) {
    case 0: return q->res_down->num_halfband_stages;
    case 1: return q->res_up->num_halfband_stages;
    case 2: return q->res_up->arbitrary->step;
    case 3: return q->res_down->arbitrary->step;
    default:
        if (what >= 4 && (unsigned)(what - 4) < q->res_up->num_halfband_stages) return q->res_up->m_stage[what - 4];
        return 0;
    }
}

unsigned orc_dsd_design(const orc_dsd *q, int what, float *out, unsigned cap)
{
    if (!q) return 0;
    const float *src = NULL; unsigned n = 0;
    if (what == 0) { src = q->res_up->arbitrary->bank; n = q->res_up->arbitrary->npfb * q->res_up->arbitrary->sub_len; }
    else if (what >= 1 && (unsigned)(what - 1) < q->res_up->num_halfband_stages) {
        src = q->res_up->stage[what - 1]->h1; n = q->res_up->stage[what - 1]->h1_len;
    }
    if (out && src) memcpy(out, src, (size_t)(n < cap ? n : cap) * sizeof(float));
    return n;
}

int orc_dsd_process_block(orc_dsd *q, const cf32 *iq, unsigned n_in, int16_t *pcm, float *audio, unsigned cap,
                          unsigned *n_out, cf32 *resampled, float *fm, unsigned *n_resampled)
{
    if (!q || n_in > q->cfg.max_block || (n_in && !iq)) return 1;
    unsigned ny = 0, nz = 0;
    orc_iirfilt_crcf_execute_block(q->dcblock, iq, n_in, q->buffp);                   /* :167 */
    orc_msresamp_crcf_execute(q->res_down, q->buffp, n_in, q->resamp_buf, &ny);       /* :168 */
    if (ny > q->res_size) return 2;
    orc_freqdem_demodulate_block(&q->fm_demod, q->resamp_buf, ny, q->fm_out_buf);     /* :169 */
    orc_msresamp_rrrf_execute(q->res_up, q->fm_out_buf, ny, q->out_buf, &nz);         /* :170 */
    if (nz > q->out_size) return 2;
    if (n_out) *n_out = nz;
    if (n_resampled) *n_resampled = ny;
    if (resampled) memcpy(resampled, q->resamp_buf, (size_t)ny * sizeof(cf32));
    if (fm) memcpy(fm, q->fm_out_buf, (size_t)ny * sizeof(float));
    if ((pcm || audio) && nz > cap) return 3;
    /* :172-175  buf_out_s[i] = out_buf[i] * INT16_MAX  (float -> int16 truncates toward zero; saturation is this
     * build's addition, the C conversion is undefined outside the int16 range).  NOTE the reference declares
     * buf_out_s[res_size] (:145) but writes nz ~ 3.84 ny samples into it -- a stack overflow for full chunks that the
     * restatement does not reproduce. */
    for (unsigned i = 0; i < nz; i++) {
        if (audio) audio[i] = q->out_buf[i];
        if (pcm) pcm[i] = orc_pcm_from_float(q->out_buf[i]);
    }
    return 0;
}
