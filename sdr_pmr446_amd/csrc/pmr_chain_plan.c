/* pmr_chain_plan.c -- design-time side of the handle: the plan (which kernels, tile geometry, tables uploaded once), the closed-form
 * block accounting, and the host-only pmr_cfg_* helpers.  Reference: the set-up code of src/sdr_pmr446.c:420-468. */
#include "pmr_chain_priv.h"


/* Tables of the FFT form of the audio FIR for the folded tap set g[n] (pmr_fir_fft.hip): spectra in the kernel's position order and
 * exact twiddles, both transform sizes.  h2 != NULL: the second tap set of the DUAL pass (CTCSS low-pass branch), zero-extended to n. */
int fir_fft_upload_spectrum(pmr_chain q, float **dst, unsigned N, const float *h, unsigned n)
{
    float *H = (float *)malloc((size_t)N * 2 * sizeof(float));
    if (!H) return fail(q, PMR_ENOMEM, "malloc", hipSuccess);
    pmr_fir_fft_spectrum(N, h, n, H);
    const int rc = dev_upload(q, dst, H, (size_t)N * 2);
    free(H);
    return rc;
}

static int fir_fft_init(pmr_chain q, const float *g, unsigned n)
{
    q->fft_ok = 0;
    if (q->sw.fir_direct || q->cfg.deemph_fir || q->cfg.lowpass || !pmr_fir_fft_supported(q->M, n))
        return PMR_OK;
    for (int w = 0; w < 3; w++) {
        const unsigned N = pmr_fir_fft_size(w);
        int rc = fir_fft_upload_spectrum(q, &q->d_fft_H[w], N, g, n);
        if (rc) return rc;
        float *ta = (float *)malloc((size_t)15 * (N / 16) * 2 * sizeof(float)), *tbv = (float *)malloc((size_t)(N / 256) * 16 * 2 * sizeof(float));
        if (!ta || !tbv) { free(ta); free(tbv); return fail(q, PMR_ENOMEM, "malloc", hipSuccess); }
        pmr_fir_fft_twiddles(N, ta, tbv);
        rc = dev_upload(q, &q->d_fft_TA[w], ta, (size_t)15 * (N / 16) * 2);
        if (!rc) rc = dev_upload(q, &q->d_fft_TB[w], tbv, (size_t)(N / 256) * 16 * 2);
        free(ta); free(tbv);
        if (rc) return rc;
        q->fft_tab[w].H = q->d_fft_H[w]; q->fft_tab[w].H2 = NULL; q->fft_tab[w].TA = q->d_fft_TA[w]; q->fft_tab[w].TB = q->d_fft_TB[w];
    }
    q->fft_ok = 1;
    return PMR_OK;
}

/* Which form runs the audio FIR of this block?  The FFT form where the block is large (>= 2^17 output samples: every 2^22+-sample
 * block of cfg2 / cfg3 / cfg5, all channels or one); small blocks (the reference's 100 000-sample calls: 1220 frames x 16) keep the
 * direct MFMA form and its RSSI rider.  Transform size: 1024 points -- one-wave workgroups with 8.7 KB of LDS that fit beside the
 * front end's tiles.  The 4096-point form does 0.7x the arithmetic (92 % instead of 63 % of a block is output) and is the faster
 * kernel alone at cfg3, but in the chain it measured 4 % slower at cfg2 and equal at cfg3 (profiles/r04_ab_log.txt): it is
 * compiled in and selected by PMR_FIR=fft4096 only (a run-time switch, read at create: tests/test_gpu_fir_fft.py runs both sizes;
 * round 4's compile-time threshold lived in this C file, where the tools' -D flags for hipcc never reached it).
 * Returns -1 (direct), 0 (1024 points) or 1 (4096 points). */
int fir_fft_pick(const struct pmr_chain_s *q, unsigned ns, unsigned nchan, int dual)
{
    if (!q->fft_ok || (unsigned long long)ns * nchan < (1ull << 17)) return -1;
    /* Transform size: 2048 points (two-wave workgroups, 18.4 KB of LDS: still fits beside four front-end tiles) where that takes at
     * least 15 % fewer transform points than 1024 (82 % instead of 63 % of a block is output): every 2^22+-sample block of cfg2 / cfg3;
     * 1024 points where the call's frame count leaves a 2048-point block half empty (cfg5: 838 frames = two 1024-point blocks or ONE
     * 2048-point block).  Six interleaved repetitions on one box (profiles/r05_ab_log.txt r5i): cfg2 452.4 vs 444.1 GS/s (+1.9 %, 6 of 6),
     * cfg3 456.3 vs 453.9 (+0.5 %); cfg5 with 2048 points forced: -1.5 %.  PMR_FIR=fft1024 / fft2048 / fft4096 force a size. */
    int which = 0;
    {
        const unsigned long long T = q->hp_len - 1u;
        const unsigned long long n0 = ((unsigned long long)ns + (1024ull - T) - 1ull) / (1024ull - T) * 1024ull;
        const unsigned long long n2 = ((unsigned long long)ns + (2048ull - T) - 1ull) / (2048ull - T) * 2048ull;
        /* (DUAL -- the CTCSS low-pass branch as second product -- doubles the workgroup's LDS: 36.8 KB at 2048 points no longer fits
         *  beside four front-end tiles: with many open channels -2.3 % at cfg2, with ONE open channel -- a handful of workgroups, the
         *  reference's mode -- +1.6 %; measured r5i) */
        if (n2 * 100ull <= n0 * 85ull && (!dual || nchan <= 2)) which = 2;
    }
    if (q->sw.fir_fft1024) which = 0;
    if (q->sw.fir_fft2048) which = 2;
    if (q->sw.fir_fft4096) which = 1;
    /* the kernel indexes the discriminator ring and its grid with 32-bit arithmetic (pmr_launch_fir_fft re-checks the same limits):
     * a plan beyond them takes the direct form HERE, before anything is launched -- a launch-time refusal would come in the middle
     * of a block and fault the handle on every large block */
    const unsigned long long N = pmr_fir_fft_size(which), L = N - (q->hp_len - 1u);
    if ((q->fm_mask + 1ull) * q->M > 0xffffffffull || (unsigned long long)ns + N > 0x7fffffffull) return -1;
    if (((unsigned long long)ns + L - 1ull) / L * ((nchan + 1ull) / 2ull) > 0x7fffffffull) return -1;
    return which;
}

/* ------------------------------------------------------------------------------------------- */

void pmr_chain_default_cfg(pmr_chain_cfg *c)
{
    memset(c, 0, sizeof(*c));
    c->fs_in = 1024000.0;            /* include/sdr_pmr446.h:13 */
    c->num_channels = 16;            /* src/sdr_pmr446.c:23 */
    c->channel_width_hz = 12500.0;   /* :22 */
    c->dcblock_alpha = 0.0005f;      /* :422 */
    c->resamp_As = 60.0f;            /* :426 */
    c->pfb_m = 13; c->pfb_As = 80.0f; /* :437 */
    c->fm_kf = 0.5f;                 /* :440 */
    c->audio_gain = 4.0f;            /* :33 */
    c->lowpass = 0;                  /* :154 */
    c->deemph_fir = 0;               /* :457 */
    c->max_block = 100000;           /* :30 */
    c->device = -1;
}

/* ------------------------------------------------------------------------------------------- */
/* fused front end: tile geometry and the closed-form gains of the cascade for an exponential     */

/* Does the plan's cascade get the two-level front end?  (Deep cascades: with 4096-sample tiles the halo would eat the tile.) */
static int fe_wants_two_levels(const pmr_design *d)
{
    const unsigned h = d->num_stages, D = d->decim;
    if (h > PMR_FE_MAX_STAGES || h < 4) return 0;
    unsigned long S = 0;
    for (unsigned e = 0; e < h; e++) S += (unsigned long)(4 * d->m_stage[h - 1 - e] - 2) << e;
    const unsigned long H = S + 13ul * D;
    if (!(h >= 5 && (4096ul - (H < 4096ul ? H : 4096ul)) * 4 < 4096ul * 3)) return 0;
    for (unsigned e = 0; e + 2 < h; e++) if (d->m_stage[h - 1 - e] != 3) return 0;
    return 1;
}

static int fe_init(pmr_chain q)
{
    const pmr_design *d = &q->d;
    const unsigned h = d->num_stages, D = d->decim;
    int rc;
    q->fe_on = 0;
    if (h > PMR_FE_MAX_STAGES) return PMR_OK;

    /* raw-sample history the cascade needs: S = sum_e (4 m_e - 2) 2^e (execution order) + 13 decimated samples */
    unsigned long S = 0;
    for (unsigned e = 0; e < h; e++) {
        q->fe_m[e] = (int)d->m_stage[h - 1 - e];
        S += (unsigned long)(4 * q->fe_m[e] - 2) << e;
    }
    unsigned long H = S + 13ul * D;
    unsigned long L = D > 16 ? D : 16;
    /* Deep cascades: with 4096-sample tiles the halo H would eat the tile.  Split: level 1 = dc-block + all but the last two
     * stages (6-tap filters, halo 10*(2^s1 - 1) raw samples) -> decimated ring; level 2 = the m = 5 and m = 10 stages +
     * resampler on the 2^s1-times decimated stream.  Costs 16/2^s1 B per raw sample of extra HBM traffic (1 B at s1 = 4). */
    q->fe_two = 0; q->fe_s1 = 0;
    if (fe_wants_two_levels(d)) { q->fe_two = 1; q->fe_s1 = (int)h - 2; }
    const unsigned s1 = (unsigned)q->fe_s1, D1 = 1u << s1;
    if (q->fe_two) {                              /* level-1 geometry replaces the single-level one below */
        S = 0;
        for (unsigned e = 0; e < s1; e++) S += (unsigned long)(4 * q->fe_m[e] - 2) << e;
        H = S;
#ifdef EXP_L1_EXTRA_HALO   /* experiment (results stay CORRECT: a larger halo only): what would the halo of a level 1 that also ran the first
                            * long stage cost?  (4 m - 2) 2^s1 = 288 more raw samples per tile at m = 5, s1 = 4 -- profiles/r06_ab_log.txt r6b */
        H += EXP_L1_EXTRA_HALO;
#endif
        L = D1 > 16 ? D1 : 16;
    }
    int nt = 0;
    unsigned long T_own = 0;
    {
        /* tile geometries (threads x 16 samples): 256 -> 4096-sample tiles; 1024 -> 16384 (cascades too deep for those) */
        const int cands[2] = { 256, 1024 };
        for (int ci = 0; ci < 2 && !nt; ci++) {
            const unsigned long N0c = (unsigned long)cands[ci] * 16;
            if (N0c % L || H + L > N0c) continue;
            const unsigned long t = (N0c - H) / L * L;
            if ((cands[ci] == 256 && t * 4 >= N0c * 3) || (cands[ci] == 1024 && t * 2 >= N0c)) { nt = cands[ci]; T_own = t; }
        }
    }
    if (!nt) return PMR_OK;                      /* cascade too deep for one LDS tile: staged path */
    const unsigned long N0 = (unsigned long)nt * 16;
    q->fe_nt = nt; q->fe_spt = 16;
    q->fe_T_own = (int)T_own;
    const unsigned Dl = q->fe_two ? D1 : D;       /* decimation of the (first) level */
    q->fe_Hh = (int)(N0 - T_own);
    q->fe_HhQ = q->fe_Hh / (int)Dl;
    q->fe_TQ = (int)(T_own / Dl);
    q->fe_hcap = (int)((q->fe_Hh + Dl + 15) / 16 * 16);
    q->fe_max_tiles = (unsigned)((q->cfg.max_block + Dl) / T_own + 2);
    if (q->fe_two) {
        if (nt != 256) return PMR_OK;             /* level kernels exist for 256 x 16 tiles only */
        const unsigned D2 = 1u << (h - s1);
        unsigned long S2 = 0;
        for (unsigned e = s1; e < h; e++) S2 += (unsigned long)(4 * q->fe_m[e] - 2) << (e - s1);
        const unsigned long H2 = S2 + 13ul * D2, L2 = D2 > 16 ? D2 : 16;
        /* level-2 tile: 2048 ring samples for the specialised kernel (k_fe_level2<MA, MB>), 4096 for the generic one */
        q->fe2_fast = h - s1 == 2 && pmr_fe_fast_covers(2, q->fe_m + s1, 2);      /* (MA, MB) is one of the pairs k_fe_level2 is built for */
        const unsigned long N2 = q->fe2_fast ? 2048 : 4096;
        if (H2 + L2 > N2) return PMR_OK;
        const unsigned long t2 = (N2 - H2) / L2 * L2;
        q->fe2_N0 = (int)N2;
        q->fe2_T_own = (int)t2; q->fe2_Hh = (int)(N2 - t2); q->fe2_HhQ = q->fe2_Hh / (int)D2; q->fe2_TQ = (int)(t2 / D2);
        uint64_t need = (uint64_t)q->fe2_Hh + D2 + (uint64_t)PIPE_DEPTH * ((q->cfg.max_block >> s1) + 2) + 64, cap = 1;
        while (cap < need) cap <<= 1;
        q->ring1_mask = cap - 1;
        if ((rc = dev_alloc_state(q, (void **)&q->d_fe_ring1, (size_t)cap * sizeof(cfl)))) return rc;
    }

    /* branch taps of all stages, execution order */
    {
        float tmp[PMR_FE_MAX_STAGES * 64];
        int off = 0;
        for (unsigned e = 0; e < h; e++) {
            const unsigned g = h - 1 - e, n = 2 * d->m_stage[g];
            if (off + n > sizeof(tmp) / sizeof(tmp[0])) return PMR_OK;
            q->fe_tap_off[e] = off;
            memcpy(tmp + off, d->hb_h1[g], n * sizeof(float));
            off += (int)n;
        }
        memcpy(q->fe_taps_host, tmp, sizeof(tmp));
        if ((rc = dev_upload(q, &q->d_fe_taps, tmp, off ? off : 1))) return rc;
    }

    /* gains for yb_err[r] = alpha V lambda^r:  stage e maps A mu^n -> A G_e (mu^2)^i with
     * G_e = mu * sum_k hb_e[k] mu^-k;  after the cascade dec_err[q'] = alpha zeta prod(G_e) V mu_h^q';
     * the arbitrary resampler adds GA[idx] = sum_n hA[idx + 256 n] mu_h^-n.  All in double.          */
    const double lam = d->dc_lambda, alpha = 1.0 - lam;
    double mu = lam, G = 1.0, mu1 = lam;
    q->fe1_K = (float)alpha;
    for (unsigned e = 0; e < h; e++) {
        const unsigned g = h - 1 - e, n = 4 * d->m_stage[g] + 1;
        double acc = 0.0;
        for (unsigned k = 0; k < n; k++) acc += (double)d->hb_proto[g][k] * pow(mu, -(double)k);
        G *= mu * acc;
        mu *= mu;
        if (q->fe_two && e + 1 == s1) { q->fe1_K = (float)(alpha * G); mu1 = mu; }
    }
    q->fe_Kgain = (float)(alpha * (double)d->zeta * G);
    if (q->fe_two) mu = mu1;                      /* the carry is removed at the level-1 output: tables for mu_s1 */
    {
        float ga[PMR_ARB_NPFB];
        for (unsigned idx = 0; idx < PMR_ARB_NPFB; idx++) {
            double acc = 0.0;
            for (unsigned n = 0; n < 2 * PMR_ARB_M; n++)
                acc += (double)d->arb_proto[idx + PMR_ARB_NPFB * n] * pow(mu, -(double)n);
            ga[idx] = (float)acc;
        }
        if ((rc = dev_upload(q, &q->d_fe_GA, ga, PMR_ARB_NPFB))) return rc;
        {
            /* one-level form: the carry's gain per polyphase branch with the cascade's gain folded in, Kgain * GA[idx] as ONE float
             * product: what k_fe_tilefix, k_fe_carry_tail and the channelizers' loads all multiply by mu^q' */
            float gak[PMR_ARB_NPFB];
            for (unsigned idx = 0; idx < PMR_ARB_NPFB; idx++) gak[idx] = q->fe_Kgain * ga[idx];
            if ((rc = dev_upload(q, &q->d_fe_GAK, gak, PMR_ARB_NPFB))) return rc;
        }
        const unsigned nq = (unsigned)(N0 / (q->fe_two ? D1 : D)), n1 = nq / 32 + 2;
        float *t1 = (float *)calloc(n1, sizeof(float)), t2[32];
        if (!t1) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
        for (unsigned i = 0; i < n1; i++) t1[i] = (float)pow(mu, 32.0 * i);
        for (unsigned i = 0; i < 32; i++) t2[i] = (float)pow(mu, (double)i);
        rc = dev_upload(q, &q->d_fe_T1, t1, n1);
        if (!rc && q->fe_two) {
            /* level 2 applies level 1's carry while loading: one table of the whole gain K1 * mu^e, the SAME float products
             * k_fe_carry forms from T1 / T2 (K * (T1[e >> 5] * T2[e & 31])), so both correct a sample identically */
            float *g1 = (float *)calloc((size_t)n1 * 32, sizeof(float));
            if (!g1) rc = fail(q, PMR_ENOMEM, "calloc", hipSuccess);
            else {
                for (unsigned e = 0; e < n1 * 32; e++) { const float tt = t1[e >> 5] * t2[e & 31]; g1[e] = q->fe1_K * tt; }
                rc = dev_upload(q, &q->d_fe_G1, g1, (size_t)n1 * 32);
                free(g1);
            }
        }
        if (!rc && !q->fe_two) {
            /* one-level form, carry applied at the channelizer's loads: mu^q' as ONE table holding the float products
             * T1[q' >> 5] * T2[q' & 31] that k_fe_tilefix forms, so both correct a sample identically */
            float *g12 = (float *)calloc((size_t)n1 * 32, sizeof(float));
            if (!g12) rc = fail(q, PMR_ENOMEM, "calloc", hipSuccess);
            else {
                for (unsigned e = 0; e < n1 * 32; e++) g12[e] = t1[e >> 5] * t2[e & 31];
                rc = dev_upload(q, &q->d_fe_G12, g12, (size_t)n1 * 32);
                free(g12);
            }
        }
        free(t1);
        if (rc) return rc;
        if ((rc = dev_upload(q, &q->d_fe_T2, t2, 32))) return rc;
    }
    {
        float ll[72];
        const double spt = (double)q->fe_spt;
        for (unsigned l = 0; l < 72; l++) ll[l] = (float)pow(lam, spt * l);
        if ((rc = dev_upload(q, &q->d_fe_lam_lane, ll, 72))) return rc;
        for (int j = 0; j < 6; j++) q->fe_lam_pow16[j] = (float)pow(lam, spt * (double)(1u << j));
        q->fe_lam_wave = (float)pow(lam, 64.0 * spt);
    }
    for (int i = 0; i < 2; i++) {
        if ((rc = dev_alloc_state(q, (void **)&q->d_fe_hist[i], (size_t)q->fe_hcap * sizeof(cfl)))) return rc;
        if ((rc = dev_alloc_state(q, (void **)&q->d_fe_vstate[i], sizeof(cfl)))) return rc;
    }
    /* probes / tile ranges: one set per block in flight (the carry kernel of block b runs on the back-end stream while the
     * front end of block b+1 is already writing its own) */
    if ((rc = dev_alloc(q, (void **)&q->d_fe_probeA, (size_t)PIPE_DEPTH * q->fe_max_tiles * sizeof(cfl)))) return rc;
    if ((rc = dev_alloc(q, (void **)&q->d_fe_probeB, (size_t)PIPE_DEPTH * q->fe_max_tiles * sizeof(cfl)))) return rc;
    for (unsigned i = 0; i < PIPE_DEPTH; i++)
        if ((rc = dev_alloc(q, (void **)&q->d_fe_V[i], (size_t)q->fe_max_tiles * sizeof(cfl)))) return rc;
    if ((rc = dev_alloc(q, (void **)&q->d_fe_probeL, PIPE_DEPTH * sizeof(cfl)))) return rc;
    if ((rc = dev_alloc(q, (void **)&q->d_fe_probeE, PIPE_DEPTH * sizeof(cfl)))) return rc;
    if ((rc = dev_alloc(q, (void **)&q->d_fe_tile_j, (size_t)PIPE_DEPTH * q->fe_max_tiles * 2 * sizeof(uint64_t)))) return rc;
    {
        /* carry look-back length and the powers of rho = lambda^T_own it needs */
        const double rho = pow(lam, (double)T_own);
        double kterms = rho > 0.0 && rho < 1.0 ? ceil(log(1e-12) / log(rho)) : 1.0;
        if (kterms < 1.0) kterms = 1.0;
        if (kterms > 4096.0) kterms = 4096.0;
        q->fe_K = (unsigned)kterms;
        float *rp = (float *)calloc(q->fe_K + 2, sizeof(float));
        if (!rp) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
        for (unsigned k = 0; k <= q->fe_K + 1; k++) rp[k] = (float)pow(rho, (double)k);
        rc = dev_upload(q, &q->d_fe_rho_pow, rp, q->fe_K + 2);
        free(rp);
        if (rc) return rc;
    }
    q->fe_sel = 0;
    q->fe_on = 1;
    q->fe_fast_fmt = nt == 256 &&
                     pmr_fe_fast_covers(q->fe_two ? 1 : 0, q->fe_m, q->fe_two ? q->fe_s1 : (int)h);
    return PMR_OK;
}

int chain_init(pmr_chain q)
{
    const pmr_design *d = &q->d;
    const unsigned M = q->M, p = d->pfb_p, h = d->num_stages;
    int rc;

    /* constant tables */
    for (unsigned g = 0; g < h; g++)
        if ((rc = dev_upload(q, &q->d_hb_h1[g], d->hb_h1[g], 2 * d->m_stage[g]))) return rc;
    if ((rc = dev_upload(q, &q->d_arb_bank, d->arb_bank, (size_t)PMR_ARB_NPFB * 2 * PMR_ARB_M))) return rc;
    if ((rc = dev_upload(q, &q->d_pfb_taps_t, d->pfb_taps_t, (size_t)p * M))) return rc;
    if ((rc = dev_upload(q, &q->d_fft_tw, d->fft_tw, M))) return rc;
    if ((rc = dev_upload(q, &q->d_nco_cs, d->nco_cs, (size_t)d->nco_period * 2))) return rc;

    /* dc-block scan constants, evaluated in double */
    q->dcc.a1 = d->dc_a1;
    for (int j = 0; j < 8; j++) q->dcc.lam_pow16[j] = (float)pow(d->dc_lambda, 16.0 * (double)(1u << j));
    {
        double lt = pow(d->dc_lambda, (double)PMR_DC_TILE);
        for (int j = 0; j < 10; j++) q->dcc.lam_tile_pow[j] = (float)pow(lt, (double)(1u << j));
        float tmp[1024];
        for (unsigned t = 0; t < 256; t++) tmp[t] = (float)pow(d->dc_lambda, 16.0 * t);
        if ((rc = dev_upload(q, &q->d_lam_thread_pow, tmp, 256))) return rc;
        for (unsigned t = 0; t < 1024; t++) tmp[t] = (float)pow(lt, (double)t);
        if ((rc = dev_upload(q, &q->d_lam_tile_idx_pow, tmp, 1024))) return rc;
    }

    /* audio filter tables (:443-458); NULL selects the PMR446 tables of :56-136 */
    const float *hp = q->cfg.hp_taps ? q->cfg.hp_taps : pmr446_hp_audio_taps;
    const float *lp = q->cfg.lp_taps ? q->cfg.lp_taps : pmr446_lp_audio_taps;
    const float *de = q->cfg.deemph_taps ? q->cfg.deemph_taps : pmr446_deemph_taps;
    q->hp_len = q->cfg.hp_taps ? q->cfg.hp_len : PMR446_HP_AUDIO_TAPS_LEN;
    q->lp_len = q->cfg.lp_taps ? q->cfg.lp_len : PMR446_LP_AUDIO_TAPS_LEN;
    q->de_len = q->cfg.deemph_taps ? q->cfg.deemph_len : PMR446_DEEMPH_TAPS_LEN;
    if (q->hp_len < 1 || q->hp_len + PMR_AUDIO_J > FM_HIST_FRAMES || q->lp_len < 1 || q->de_len < 1 ||
        q->lp_len + PMR_AUDIO_J > AUX_HIST_FRAMES || q->de_len + PMR_AUDIO_J > AUX_HIST_FRAMES)
        return fail(q, PMR_EINVAL, "audio filter length out of range", hipSuccess);
    {
        /* The audio kernels run ONE FIR: gain (:890) and, for the default IIR de-emphasis (:898), its impulse
         * response e[0] = b0, e[k] = (b1 - a1 b0)(-a1)^(k-1) are folded into the high-pass taps in double.  The pole is
         * 0.0146, so 7 terms reproduce the recursion to ~2e-11 (checked against scipy.lfilter); no per-thread IIR warm-up. */
        const unsigned ke = q->cfg.deemph_fir ? 1 : 7, n = q->hp_len + ke - 1;
        double e[8] = {0};
        if (q->cfg.deemph_fir) e[0] = 1.0;
        else {
            const double b0 = d->de_b0, b1 = d->de_b1, a1 = d->de_a1;
            e[0] = b0;
            for (unsigned k = 1; k < ke; k++) e[k] = (b1 - a1 * b0) * pow(-a1, (double)(k - 1));
        }
        float *g = (float *)calloc(n, sizeof(float));
        if (!g) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
        for (unsigned i = 0; i < n; i++) {
            double acc = 0.0;
            for (unsigned k = 0; k < ke && k <= i; k++)
                if (i - k < q->hp_len) acc += e[k] * (double)hp[i - k];
            g[i] = (float)((double)q->cfg.audio_gain * acc);
        }
        rc = upload_padded_taps(q, &q->d_hp_pad, g, n);
        if (!rc) rc = fir_fft_init(q, g, n);
        free(g);
        if (rc) return rc;
        q->hp_len_raw = q->hp_len;
        q->hp_len = n;
    }
    if ((rc = upload_padded_taps(q, &q->d_lp_pad, lp, q->lp_len))) return rc;
    if ((rc = upload_padded_taps(q, &q->d_de_pad, de, q->de_len))) return rc;

    /* state + work buffers */
    const unsigned mb = q->cfg.max_block;
    if ((rc = dev_alloc(q, (void **)&q->d_in, (size_t)mb * sizeof(cfl)))) return rc;
    if ((rc = dev_alloc_state(q, (void **)&q->d_dc_state, sizeof(cfl)))) return rc;
    const unsigned max_tiles = (mb + PMR_DC_TILE - 1) / PMR_DC_TILE + 1;
    if ((rc = dev_alloc(q, (void **)&q->d_dc_agg, (size_t)max_tiles * sizeof(cfl)))) return rc;
    if ((rc = dev_alloc(q, (void **)&q->d_dc_W, (size_t)max_tiles * sizeof(cfl)))) return rc;
    for (unsigned e = 0; e <= h; e++) {
        /* stage e (execution order) is design stage h-1-e; z_h feeds the arbitrary resampler */
        q->keep[e] = e < h ? 4 * d->m_stage[h - 1 - e] : ARB_KEEP;
        size_t cap = (size_t)q->keep[e] + ((size_t)mb >> e) + 2;
        if ((rc = dev_alloc_state(q, (void **)&q->d_z[e], cap * sizeof(cfl)))) return rc;
    }
    /* rings sized for the filter history plus PIPE_DEPTH blocks, so block b+1's front end never overwrites what block b's
     * back end still reads */
    {
        uint64_t need = (uint64_t)(p + 1) * M + (uint64_t)PIPE_DEPTH * q->res_size + 64, cap = 1;
        while (cap < need) cap <<= 1;
        q->xr_mask = cap - 1;
        if ((rc = dev_alloc_state(q, (void **)&q->d_xr, (size_t)cap * sizeof(cfl)))) return rc;
        need = (uint64_t)FM_HIST_FRAMES + (uint64_t)PIPE_DEPTH * q->chan_size + 64; cap = 1;
        while (cap < need) cap <<= 1;
        q->fm_mask = cap - 1;
        if ((rc = dev_alloc_state(q, (void **)&q->d_fm, (size_t)cap * M * sizeof(float)))) return rc;
        if (q->cfg.deemph_fir || q->cfg.lowpass) {
            if ((rc = dev_alloc_state(q, (void **)&q->d_aux1, (size_t)cap * M * sizeof(float)))) return rc;
            if ((rc = dev_alloc_state(q, (void **)&q->d_aux2, (size_t)cap * M * sizeof(float)))) return rc;
        }
    }
    q->scratch_bytes = 4096;         /* history shifts of the staged front end only (<= 40 samples each) */
    if ((rc = dev_alloc(q, &q->d_scratch, q->scratch_bytes))) return rc;
    q->rssi_part_cap = ((size_t)q->chan_size + 2) * M;   /* worst case: one new frame per channelizer tile */
    if ((rc = dev_alloc(q, (void **)&q->d_rssi_part, q->rssi_part_cap * sizeof(float)))) return rc;

    if ((rc = dev_alloc(q, (void **)&q->d_chan_list, (size_t)M * sizeof(unsigned)))) return rc;
    if ((rc = dev_alloc(q, (void **)&q->d_reset_flags, M))) return rc;
    if (!(q->h_reset_flags = (uint8_t *)calloc(M, 1))) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
    if (!(q->h_open = (uint8_t *)malloc(M))) return fail(q, PMR_ENOMEM, "malloc", hipSuccess);
    memset(q->h_open, 1, M);
    q->n_enabled = M; q->mask_on = 0; q->reset_pending = 0;

    if ((rc = fe_init(q))) return rc;
    q->chan_small = pmr_channelize_small_supported(M, p, d->nco_period);
    q->chan_wide = !q->chan_small && pmr_channelize_wide_supported(M, p, d->nco_period);
    if (q->chan_wide && (rc = dev_alloc(q, (void **)&q->d_chan_x, ((size_t)q->chan_size + 2) * M * sizeof(cfl)))) return rc;
    q->l2_on_backend = 1;
    /* 256-channel one-level plans (cfg3, every GPU of cfg4): 6.5 KB of unused LDS per front-end tile -- three 40 KB tiles per CU instead
     * of four 33.6 KB ones, 40 KB of every CU left to the back end, whose 256-channel bank now takes 56 KB per workgroup (24 frames).
     * Chain +3.3 / +2.4 / +1.6 % on three boxes (459.7 vs 445.5, 444.8 vs 434.8, 458.2 vs 451.0 GS/s; 5 KB of padding, which still
     * lets four tiles in, measures the same); alone the kernel is 1.5 % slower.  The same padding COSTS cfg2 2.6 % and cfg5 2.3 %.
     * profiles/r04_ab_log.txt r4r. */
#ifndef FE_LDS_PAD_256
#define FE_LDS_PAD_256 6656u      /* (sweep hook: tools/ab_libs.py builds) */
#endif
    q->fe_lds_pad = (q->fe_on && !q->fe_two && q->chan_wide && M == 256) ? FE_LDS_PAD_256 : 0u;
#ifdef FE_LDS_PAD_16        /* sweep hook: the same padding for the 16-channel plan (r6i) */
    if (q->fe_on && !q->fe_two && q->chan_small) q->fe_lds_pad = FE_LDS_PAD_16;
#endif
    q->tf_on_backend = 0;
    q->cal_ok = 0;
    if (q->fe_on && !q->fe_two && q->d_fe_G12 && !q->sw.carry_inplace) {
        q->cal_adv_q = (unsigned)(((uint64_t)M * d->arb_step) >> 24) + 1u;
        q->cal_ok = pmr_channelize_carry_at_load(M, p, d->nco_period, q->chan_small, q->chan_wide,
                                                 q->cal_adv_q, (unsigned)q->fe_TQ);
        q->cal_nv = pmr_channelize_carry_nv(M, q->cal_adv_q, (unsigned)q->fe_TQ);
        q->cal_nbias = (unsigned)(((uint64_t)(p + 4) * q->cal_adv_q) / (unsigned)q->fe_TQ) + 2u;
    }

    q->n_raw = 0; q->arb_phase = 0; q->xr_abs = 0; q->frames_done = 0; q->n_calls = 0;
    HIPCHK(hipStreamSynchronize(q->stream), "init sync");
    return PMR_OK;
}

/* Closed-form sample accounting for a block of n_in raw samples (no device round trip):
 *   decimated samples Q = floor((n_raw+n_in)/D) - floor(n_raw/D)      (msresamp buffer_index rule)
 *   resampled outputs ny from the 24-bit phase accumulator            (resamp_crcf)
 *   frames ns = floor((leftover + ny) / M)                            (ring rule, :804)             */
void plan_core(unsigned num_stages, uint32_t arb_step, unsigned M, uint64_t n_raw, uint32_t arb_phase,
                      unsigned leftover, unsigned n_in, unsigned *ny_out, unsigned *ns_out, uint32_t *phase_out)
{
    const uint64_t Q = ((n_raw + n_in) >> num_stages) - (n_raw >> num_stages);
    const uint64_t span = Q << 24;
    unsigned ny = 0;
    if (Q && (uint64_t)arb_phase < span) ny = (unsigned)((span - arb_phase + arb_step - 1) / arb_step);
    *ny_out = ny;
    *ns_out = (leftover + ny) / M;
    if (phase_out) *phase_out = (uint32_t)((uint64_t)arb_phase + (uint64_t)ny * arb_step - span);
}

void plan_counts(const struct pmr_chain_s *q, unsigned n_in, unsigned *ny_out, unsigned *ns_out)
{
    plan_core(q->d.num_stages, q->d.arb_step, q->M, q->n_raw, q->arb_phase,
              (unsigned)(q->xr_abs - q->frames_done * q->M), n_in,
              ny_out, ns_out, NULL);
}

/* ---- host-only helpers (no device) ---- */
static int cfg_design(const pmr_chain_cfg *cfg, pmr_design *d)
{
    if (!cfg) return 1;
    return pmr_design_build(d, cfg->fs_in, cfg->num_channels, cfg->channel_width_hz, cfg->dcblock_alpha,
                            cfg->resamp_As, cfg->pfb_m, cfg->pfb_As, cfg->fm_kf);
}

unsigned pmr_cfg_info(const pmr_chain_cfg *cfg, int what, unsigned idx)
{
    struct pmr_chain_s tmp;
    memset(&tmp, 0, sizeof(tmp));
    if (cfg_design(cfg, &tmp.d)) { pmr_design_free(&tmp.d); return 0; }
    tmp.M = cfg->num_channels;
    unsigned r = pmr_chain_info(&tmp, what, idx);
    pmr_design_free(&tmp.d);
    return r;
}

unsigned pmr_cfg_design(const pmr_chain_cfg *cfg, int what, unsigned idx, float *out, unsigned cap)
{
    struct pmr_chain_s tmp;
    memset(&tmp, 0, sizeof(tmp));
    if (cfg_design(cfg, &tmp.d)) { pmr_design_free(&tmp.d); return 0; }
    tmp.M = cfg->num_channels;
    unsigned r = pmr_chain_design(&tmp, what, idx, out, cap);
    pmr_design_free(&tmp.d);
    return r;
}

unsigned pmr_cfg_max_frames(const pmr_chain_cfg *cfg)
{
    pmr_design d;
    memset(&d, 0, sizeof(d));
    unsigned rs = 0, cs = 0;
    if (!cfg_design(cfg, &d)) pmr_design_buffer_sizes(&d, cfg->max_block, &rs, &cs);
    pmr_design_free(&d);
    return cs;
}

int pmr_cfg_plan_block(const pmr_chain_cfg *cfg, pmr_plan_state *st, unsigned n_in, unsigned *ny, unsigned *ns)
{
    pmr_design d;
    memset(&d, 0, sizeof(d));
    if (!st || cfg_design(cfg, &d)) { pmr_design_free(&d); return PMR_EINVAL; }
    unsigned ny_ = 0, ns_ = 0; uint32_t ph = 0;
    plan_core(d.num_stages, d.arb_step, d.M, st->n_raw, st->arb_phase, st->leftover, n_in, &ny_, &ns_, &ph);
    st->n_raw += n_in;
    st->arb_phase = ph;
    st->leftover = (st->leftover + ny_) - ns_ * d.M;
    if (ny) *ny = ny_;
    if (ns) *ns = ns_;
    pmr_design_free(&d);
    return PMR_OK;
}
