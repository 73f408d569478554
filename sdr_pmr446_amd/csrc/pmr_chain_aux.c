/* pmr_chain_aux.c -- what rides beside the hot path: per-launch profiling events, the CTCSS detector's host side, the
 * waterfall periodogram (f4), debug capture. */
#include "pmr_chain_priv.h"

const char *const pmr_k_names[K_COUNT] = { "k_dcblock<agg>", "k_dc_scan", "k_dcblock<apply>", "k_halfband", "k_arb",
                                        "k_channelize (fused256 / pfb_wide + fft_disc / generic)", "k_rssi_finish",
                                        "audio FIR <hp> (k_fir_fft / k_fir_mfma4 / k_fir_pair)", "audio FIR <deemph>",
                                        "audio FIR <lp>", "k_fe_fast (k_frontend)", "k_fe_carry",
                                        "k_channelize_win", "k_fe_level2", "audio FIR <ctcss_lp>",
                                        "(unused)", "k_ct_seg_agg + scan + goertzel + final", "k_fe_carry_tail / k_fe_tilefix", "k_spgram + finish" };

/* ---- profiling helpers: HIP events on the chain's stream around every launch ---- */
void prof_begin(pmr_chain q, int slot, prof_pending *pp, hipStream_t st)
{
    pp->slot = -1;
    if (!q->prof_on) return;
    if (q->prof_on >= 2) return;          /* modes >= 2: only the front-end (roofline) kernel, by events its launch carries */
    hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) {
        if (q->npool) ev[i] = q->pool[--q->npool];
        else if (hipEventCreate(&ev[i]) != hipSuccess) return;
    }
    pp->a = ev[0]; pp->b = ev[1]; pp->slot = slot;
    hipEventRecord(pp->a, st);
}

void prof_push(pmr_chain q, const prof_pending *pp)
{
    if (pp->slot < 0) return;
    if (q->npend == q->cappend) {
        unsigned nc = q->cappend ? 2 * q->cappend : 256;
        prof_pending *np = (prof_pending *)realloc(q->pend, nc * sizeof(*np));
        if (!np) return;
        q->pend = np; q->cappend = nc;
    }
    q->pend[q->npend++] = *pp;
}

void prof_end(pmr_chain q, prof_pending *pp, hipStream_t st)
{
    if (pp->slot < 0) return;
    hipEventRecord(pp->b, st);
    prof_push(q, pp);
}

/* Events a front-end launch carries itself (pmr_launch_events: no packets of their own on the stream).
 *  - profile mode m >= 2: every (m-1)-th launch of the front-end kernel takes a start/stop pair (kernel begin..end);
 *  - otherwise the launch that is the front-end stream's last of this call signals "front end done" (fe_done_ev). */
void fe_launch_events(pmr_chain q, int slot, int last_on_stream, pmr_launch_events *ev, prof_pending *pp)
{
    ev->start = ev->stop = NULL;
    pp->slot = -1;
    if (q->prof_on == 1) return;                          /* mode 1 brackets every launch with records (LAUNCH_ON) */
    if (q->prof_on >= 2 && slot == K_FE && q->prof_tick++ % (unsigned)(q->prof_on - 1) == 0) {
        hipEvent_t e[2];
        for (int i = 0; i < 2; i++) {
            if (q->npool) e[i] = q->pool[--q->npool];
            else if (hipEventCreate(&e[i]) != hipSuccess) return;
        }
        pp->a = e[0]; pp->b = e[1]; pp->slot = slot;
        ev->start = e[0]; ev->stop = e[1];
        return;
    }
    if (last_on_stream && q->fe_done_ev) { ev->stop = q->fe_done_ev; q->fe_done_used = 1; }
}

void prof_resolve(pmr_chain q)
{
    for (unsigned i = 0; i < q->npend; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(q->pend[i].b) == hipSuccess &&
            hipEventElapsedTime(&ms, q->pend[i].a, q->pend[i].b) == hipSuccess) {
            q->prof_ms[q->pend[i].slot] += ms;
            q->prof_n[q->pend[i].slot]++;
        }
        if (q->npool + 2 > q->cappool) {
            unsigned nc = q->cappool ? 2 * q->cappool : 512;
            hipEvent_t *np = (hipEvent_t *)realloc(q->pool, nc * sizeof(*np));
            if (np) { q->pool = np; q->cappool = nc; }
        }
        if (q->npool + 2 <= q->cappool) { q->pool[q->npool++] = q->pend[i].a; q->pool[q->npool++] = q->pend[i].b; }
        else { hipEventDestroy(q->pend[i].a); hipEventDestroy(q->pend[i].b); }
    }
    q->npend = 0;
}

/* ---- CTCSS branch: low-pass branch FIR -> dc-block scan -> Goertzel bank, all channels ---- */
int ctcss_run(pmr_chain q, int64_t frame0, unsigned ns, int fir_done /*the low-pass branch is already in d_ctlp*/)
{
    const unsigned M = q->M, N = PMR_CT_BLOCK;
    /* tmp1 = delay188(fm) - hp(fm) (:884-889) as one FIR with taps delta_188 - h */
    if (!fir_done) LAUNCH(K_CT_FIR, pmr_launch_fir_tm(q->stream, q->d_fm, q->fm_mask, frame0, ns, M, q->d_ct_taps, q->hp_len_raw, 1.0f, 0,
                                       0.f, 0.f, 0.f, q->d_ctlp, NULL, NULL, 0, q->mask_on ? q->d_chan_list : NULL, q->n_enabled, NULL, NULL));
    if (q->dbg_on) {                                               /* the branch before ctcss_execute's dc blocker (:889 -> :606) */
        int rc_;
        if (!q->d_dbg_ct && (rc_ = dev_alloc(q, (void **)&q->d_dbg_ct, (size_t)q->chan_size * M * sizeof(float)))) return rc_;
        if ((rc_ = ring_to_linear(q, q->d_dbg_ct, q->d_ctlp, q->fm_mask, (uint64_t)frame0, ns, (size_t)M * sizeof(float)))) return rc_;
    }
    const float a1 = -1.0f + 0.0005f;                              /* iirfilt_rrrf_create_dc_blocker(0.0005f), :450 */
    /* the detector runs for the open channels only (the reference calls ctcss_execute for active_chan, :893): a closed channel's
     * dc-blocker state stays as it was, its partial Goertzel sums restart from zero when it is opened again (:867) */
    const unsigned *sel = q->mask_on ? q->d_chan_list : NULL;
    const uint64_t f0 = (uint64_t)frame0, f1 = f0 + ns;
    const unsigned nblk = (unsigned)((f1 - 1) / N - f0 / N + 1), ncomplete = (unsigned)(f1 / N - f0 / N);
    if (ncomplete > q->ct_max_ev) return fail(q, PMR_ERANGE, "ctcss events", hipSuccess);
    const int cur = q->ct_sel, nxt = cur ^ 1;
    /* pipelined calls: the detector's four kernels run on their own stream behind this block's low-pass branch; only the next
     * block's detector (same stream) and the ring-reuse gate wait for them */
    const int async = !q->cur_single && !q->dbg_on;
    hipStream_t sct = async ? q->stream_ct : q->stream;
    if (async) {
        HIPCHK(hipEventRecord(q->ev_ctlp, q->stream), "record");
        HIPCHK(hipStreamWaitEvent(q->stream_ct, q->ev_ctlp, 0), "wait low-pass branch");
    } else if (q->ct_async_last) {                                /* the previous block's detector state comes first */
        HIPCHK(hipStreamWaitEvent(q->stream, q->ev_ct[q->ct_last_par], 0), "wait detector");
    }
    /* (k_ct_final writes every open channel's carry for the next call, zeros when the call ends on a block boundary) */
#ifndef EXP_SKIP_CT     /* timing experiment: the low-pass branch is produced, the detector's kernels never run.  WRONG results */
    LAUNCH_ON(sct, K_CT_GOERTZEL, pmr_launch_ct_detector(sct, q->d_ctlp, q->fm_mask, frame0, ns, M, N, a1, q->d_ct_lampow, q->d_ct_dcstate,
                                                 q->d_ct_agg, q->d_ct_W, q->d_ct_U, q->d_ct_coef, q->d_ct_part, q->d_ct_carry[cur],
                                                 q->d_ct_carry[nxt], q->d_ct_events, q->d_ct_restart, nblk, ncomplete, sel, q->n_enabled));
#endif
    q->ct_masked_last = q->mask_on;
    if (q->mask_on) memcpy(q->ct_open_last, q->h_open, M);
    if (async) {
        HIPCHK(hipEventRecord(q->ev_ct[q->cur_par], q->stream_ct), "record");
        q->ct_ev_used[q->cur_par] = 1; q->ct_last_par = q->cur_par;
    }
    q->ct_async_last = async;
    q->ct_sel = nxt;
    q->ct_nev_last = ncomplete;
    return PMR_OK;
}

/* ---- the waterfall line.  Window as liquid's spgram scales it (oracle/orc_dsp.h):
 * hann(i, n) * sqrt(2) / (sqrt(sum w^2 / n) * sqrt(4 n)), evaluated in float like the restatement. ---- */
int pmr_chain_spectrum_enable(pmr_chain q, unsigned nfft)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    int rc = pmr_chain_synchronize(q);
    if (rc) return rc;
    if (nfft == q->spec_nfft) return PMR_OK;
    void **bufs[] = { (void **)&q->d_spec_win, (void **)&q->d_spec_tw, (void **)&q->d_spec_part, (void **)&q->d_spec_psd };
    for (size_t i = 0; i < 4; i++) if (*bufs[i]) { hipFree(*bufs[i]); *bufs[i] = NULL; }
    q->spec_nfft = 0; q->spec_ntr_last = 0;
    if (!nfft) return PMR_OK;
    if (nfft < 8 || nfft > 1024 || (nfft & (nfft - 1))) return fail(q, PMR_EINVAL, "spectrum width: a power of two, 8..1024", hipSuccess);
    const unsigned P = 4 * nfft;
    float *w = (float *)malloc(nfft * sizeof(float)), *tw = (float *)malloc(P * sizeof(float));
    if (!w || !tw) { free(w); free(tw); return fail(q, PMR_ENOMEM, "malloc", hipSuccess); }
    float g = 0.0f;
    for (unsigned i = 0; i < nfft; i++) {
        w[i] = 0.5f - 0.5f * cosf((2.0f * (float)M_PI * (float)i) / ((float)(nfft - 1)));
        g += w[i] * w[i];
    }
    g = (float)M_SQRT2 / (sqrtf(g / (float)nfft) * sqrtf((float)P));
    for (unsigned i = 0; i < nfft; i++) w[i] *= g;
    for (unsigned k = 0; k < P / 2; k++) {
        const double a = -2.0 * M_PI * (double)k / (double)P;
        tw[2 * k] = (float)cos(a); tw[2 * k + 1] = (float)sin(a);
    }
    rc = dev_upload(q, &q->d_spec_win, w, nfft);
    if (!rc) rc = dev_upload(q, &q->d_spec_tw, tw, P);
    free(w); free(tw);
    if (!rc) rc = dev_alloc(q, (void **)&q->d_spec_part, (size_t)pmr_spgram_max_workgroups() * P * sizeof(float));
    if (!rc) rc = dev_alloc(q, (void **)&q->d_spec_psd, P * sizeof(float));
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(q->stream), "sync");
    q->spec_nfft = nfft;
    return PMR_OK;
}

int pmr_chain_spectrum_read(pmr_chain q, float *psd_db, unsigned cap, unsigned *n_transforms)
{
    if (!q || !psd_db) return PMR_EINVAL;
    if (!q->spec_nfft) return fail(q, PMR_EINVAL, "spectrum not enabled", hipSuccess);
    const unsigned P = 4 * q->spec_nfft;
    if (cap < P) return fail(q, PMR_ERANGE, "spectrum buffer", hipSuccess);
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    int rc = pmr_chain_synchronize(q);
    if (rc) return rc;
    if (n_transforms) *n_transforms = q->spec_ntr_last;
    if (!q->spec_ntr_last) { memset(psd_db, 0, P * sizeof(float)); return PMR_OK; }
    HIPCHK(hipMemcpy(psd_db, q->d_spec_psd, P * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    for (unsigned i = 0; i < P; i++) psd_db[i] = 10.0f * log10f(psd_db[i]);
    return PMR_OK;
}

/* asgramcf_execute's peak search and character mapping (levels ref + k div, k = 0..9; the reference sets -40, 2 at :476) */
int pmr_asgram_ascii(const float *psd_db, unsigned nfft, unsigned n_transforms, float ref, float div, char *ascii, float *peakval,
                     float *peakfreq)
{
    static const char lc[10] = {' ', '.', ',', '-', '+', '*', '&', 'N', 'M', '#'};
    if (!psd_db || !ascii || !nfft) return PMR_EINVAL;
    const unsigned P = 4 * nfft;
    float pv = 0.0f, pf = 0.0f;
    ascii[nfft] = 0;
    if (!n_transforms) {
        memset(ascii, ' ', nfft);
    } else {
        for (unsigned i = 0; i < P; i++) if (i == 0 || psd_db[i] > pv) { pv = psd_db[i]; pf = (float)i / (float)P - 0.5f; }
        for (unsigned i = 0; i < nfft; i++) {
            float v = 0.0f;
            for (unsigned j = 0; j < 4; j++) { const float x = psd_db[4 * i + j]; v = (j == 0 || x > v) ? x : v; }
            ascii[i] = lc[0];
            for (unsigned j = 0; j < 10; j++) if (v > ref + (float)j * div) ascii[i] = lc[j];
        }
    }
    if (peakval) *peakval = pv;
    if (peakfreq) *peakfreq = pf;
    return PMR_OK;
}

int pmr_chain_ctcss_enable(pmr_chain q, int on)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    int rc = pmr_chain_synchronize(q);
    if (rc) return rc;
    if (on && !q->d_ctlp) {
        const unsigned M = q->M, N = PMR_CT_BLOCK, n = q->hp_len_raw;
        if ((n & 1) == 0) return fail(q, PMR_EINVAL, "ctcss needs an odd-length high-pass", hipSuccess);
        /* the dc-blocker scan strings at most 256 x 24 segments together per call (k_ct_seg_scan): checked HERE, before anything is
         * allocated, not by a launch that fails in the middle of a block */
        if ((q->chan_size / N + 3) * PMR_CT_SEG > pmr_ct_max_segments())
            return fail(q, PMR_ERANGE, "ctcss: max_block yields more Goertzel blocks per call than the detector strings together", hipSuccess);
        const float *hp = q->cfg.hp_taps ? q->cfg.hp_taps : pmr446_hp_audio_taps;
        float *tc = (float *)calloc(n, sizeof(float));
        if (!tc) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
        for (unsigned i = 0; i < n; i++) tc[i] = -hp[i];
        tc[(n - 1) / 2] += 1.0f;                                   /* wdelayf((n-1)/2), :447 */
        rc = upload_padded_taps(q, &q->d_ct_taps, tc, n);
        if (!rc && q->hp_len >= n) {
            /* same taps as a filter of the folded audio filter's length (zeros behind): both products then run over one window
             * in ONE pass of the MFMA kernel (pmr_launch_fir_dual) */
            float *te = (float *)calloc(q->hp_len, sizeof(float));
            if (!te) rc = fail(q, PMR_ENOMEM, "calloc", hipSuccess);
            else {
                memcpy(te, tc, n * sizeof(float));
                rc = upload_padded_taps(q, &q->d_ct_taps_ext, te, q->hp_len);
                for (int w = 0; w < 3 && !rc && q->fft_ok; w++) {      /* the low-pass branch as the FFT form's second product */
                    rc = fir_fft_upload_spectrum(q, &q->d_fft_H2[w], pmr_fir_fft_size(w), te, q->hp_len);
                    q->fft_tab[w].H2 = q->d_fft_H2[w];
                }
                free(te);
            }
        }
        free(tc);
        if (rc) return rc;
        /* Goertzel weights U_m = sin((m+1)w)/sin(w), coef = 2cos(w) as the reference computes it (:360-361) */
        float coef[PMR_CT_TONES];
        float *U = (float *)calloc((size_t)PMR_CT_TONES * (N + 1), sizeof(float));
        if (!U) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
        for (unsigned j = 0; j < PMR_CT_TONES; j++) {
            coef[j] = 2.0f * cosf((float)((2.0 * M_PI * pmr446_ctcss_freqs[j]) / q->cfg.channel_width_hz));
            const double w = acos((double)coef[j] / 2.0);
            for (unsigned i = 0; i <= N; i++) U[(size_t)j * (N + 1) + i] = (float)(sin((double)i * w) / sin(w));
        }
        rc = dev_upload(q, &q->d_ct_U, U, (size_t)PMR_CT_TONES * (N + 1));
        free(U);
        if (rc) return rc;
        if ((rc = dev_upload(q, &q->d_ct_coef, coef, PMR_CT_TONES))) return rc;
        q->ct_max_ev = q->chan_size / N + 2;
        const size_t rows = (size_t)(q->fm_mask + 1), nch = (size_t)(q->ct_max_ev + 1) * PMR_CT_SEG;     /* segments a call can touch */
        {
            float lp_[161];
            const float a1_ = -1.0f + 0.0005f;                         /* iirfilt_rrrf_create_dc_blocker(0.0005f), :450 */
            for (unsigned i = 0; i <= 160; i++) lp_[i] = (float)pow(-(double)a1_, (double)i);
            if ((rc = dev_upload(q, &q->d_ct_lampow, lp_, 161))) return rc;
        }
        if ((rc = dev_alloc_state(q, (void **)&q->d_ctlp, rows * M * sizeof(float)))) return rc;
        if ((rc = dev_alloc(q, (void **)&q->d_ct_agg, nch * M * sizeof(float)))) return rc;
        if ((rc = dev_alloc(q, (void **)&q->d_ct_W, nch * M * sizeof(float)))) return rc;
        if ((rc = dev_alloc_state(q, (void **)&q->d_ct_dcstate, (size_t)M * sizeof(float)))) return rc;
        if ((rc = dev_alloc(q, (void **)&q->d_ct_part, (size_t)(q->ct_max_ev + 1) * PMR_CT_SEG * M * PMR_CT_TONES * 2 * sizeof(float)))) return rc;
        for (int i = 0; i < 2; i++)
            if ((rc = dev_alloc_state(q, (void **)&q->d_ct_carry[i], (size_t)M * PMR_CT_TONES * 2 * sizeof(float)))) return rc;
        if ((rc = dev_alloc(q, (void **)&q->d_ct_events, (size_t)(q->ct_max_ev + 1) * M * sizeof(pmr_ctcss_event)))) return rc;
        if ((rc = dev_alloc_state(q, (void **)&q->d_ct_restart, M))) return rc;
        if (!q->ct_open_last && !(q->ct_open_last = (uint8_t *)malloc(M))) return fail(q, PMR_ENOMEM, "malloc", hipSuccess);
        memset(q->ct_open_last, 1, M);
        HIPCHK(hipStreamSynchronize(q->stream), "ctcss init");
    }
    q->ct_on = on ? 1 : 0;
    return PMR_OK;
}

float pmr_ctcss_freq(int index) { return index >= 0 && index < (int)PMR_CT_TONES ? pmr446_ctcss_freqs[index] : 0.0f; }

int pmr_chain_ctcss_read(pmr_chain q, pmr_ctcss_event *events, unsigned cap, unsigned *n_events)
{
    if (!q || !q->d_ct_events) return PMR_EINVAL;
    int rc = pmr_chain_synchronize(q);
    if (rc) return rc;
    const unsigned n = q->ct_nev_last, M = q->M;
    if (n_events) *n_events = n;
    if (!events || !n) return PMR_OK;
    if (cap < n) return fail(q, PMR_ERANGE, "ctcss event capacity", hipSuccess);
    pmr_ctcss_event *tmp = (pmr_ctcss_event *)malloc((size_t)n * M * sizeof(*tmp));
    if (!tmp) return fail(q, PMR_ENOMEM, "malloc", hipSuccess);
    hipError_t e = hipMemcpy(tmp, q->d_ct_events, (size_t)n * M * sizeof(*tmp), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        for (unsigned b = 0; b < n; b++) for (unsigned k = 0; k < M; k++) {
            if (q->ct_masked_last && !q->ct_open_last[k]) {   /* closed WHEN THE BLOCK RAN: the detector did not run (index -1, nothing detected) */
                const pmr_ctcss_event none = { -1, 0, 0.0f, 0.0f };
                events[(size_t)k * cap + b] = none;
            } else events[(size_t)k * cap + b] = tmp[(size_t)b * M + k];
        }
    free(tmp);
    return e == hipSuccess ? PMR_OK : fail(q, PMR_EHIP, "ctcss D2H", e);
}

int pmr_chain_profile_enable(pmr_chain q, int on) { if (!q) return PMR_EINVAL; q->prof_on = on; return PMR_OK; }

int pmr_chain_profile_reset(pmr_chain q)
{
    if (!q) return PMR_EINVAL;
    hipStreamSynchronize(q->stream);
    prof_resolve(q);
    memset(q->prof_ms, 0, sizeof(q->prof_ms));
    memset(q->prof_n, 0, sizeof(q->prof_n));
    return PMR_OK;
}

unsigned pmr_chain_profile_count(pmr_chain q) { (void)q; return K_COUNT; }
const char *pmr_chain_profile_name(pmr_chain q, unsigned i) { (void)q; return i < K_COUNT ? pmr_k_names[i] : NULL; }

int pmr_chain_profile_get(pmr_chain q, unsigned i, double *total_ms, unsigned *launches)
{
    if (!q || i >= K_COUNT) return PMR_EINVAL;
    hipStreamSynchronize(q->stream);
    prof_resolve(q);
    if (total_ms) *total_ms = q->prof_ms[i];
    if (launches) *launches = q->prof_n[i];
    return PMR_OK;
}

int pmr_chain_debug_enable(pmr_chain q, int on)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (on && !q->d_dbg_xr) {
        int rc;
        if ((rc = dev_alloc(q, (void **)&q->d_dbg_xr, (size_t)q->res_size * sizeof(cfl)))) return rc;
        if ((rc = dev_alloc(q, (void **)&q->d_dbg_fm, (size_t)q->chan_size * q->M * sizeof(float)))) return rc;
    }
    q->dbg_on = on;
    return PMR_OK;
}

int pmr_chain_debug_read(pmr_chain q, int what, void *host_buf, size_t cap_bytes, size_t *n_bytes)
{
    if (!q) return PMR_EINVAL;
    if (!q->dbg_on) return fail(q, PMR_EINVAL, "debug capture not enabled", hipSuccess);
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    HIPCHK(hipStreamSynchronize(q->stream), "sync");
    const void *src = NULL; size_t n = 0;
    if (what == PMR_DEBUG_RESAMPLED) { src = q->d_dbg_xr; n = (size_t)q->last_ny * sizeof(cfl); }
    else if (what == PMR_DEBUG_FM)   { src = q->d_dbg_fm; n = (size_t)q->last_ns * q->M * sizeof(float); }
    else if (what == PMR_DEBUG_CTCSS_LP && q->d_dbg_ct) { src = q->d_dbg_ct; n = (size_t)q->last_ns * q->M * sizeof(float); }
    else return PMR_EINVAL;
    if (n_bytes) *n_bytes = n;
    if (n > cap_bytes) n = cap_bytes;
    if (n) HIPCHK(hipMemcpy(host_buf, src, n, hipMemcpyDeviceToHost), "debug D2H");
    return PMR_OK;
}
