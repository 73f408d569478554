/* pmr_chain_frontend.c -- launch sequencing of the front end (dc-block + msresamp_crcf, src/sdr_pmr446.c:795-796): staged kernels,
 * the fused tile kernel, the two-level form; and the seam pmr_dsd.c borrows it through (pmr_internal.h). */
#include "pmr_chain_priv.h"


/* keep the last `keep` elements of a [src+keep]-element buffer at its front (history for the next call) */
static int shift_front(pmr_chain q, hipStream_t st, void *buf, size_t elem, size_t src, size_t keep)
{
    if (src == 0 || keep == 0) return PMR_OK;
    char *b = (char *)buf;
    if (src >= keep) {
        HIPCHK(hipMemcpyAsync(b, b + src * elem, keep * elem, hipMemcpyDeviceToDevice, st), "shift");
    } else {
        if (keep * elem > q->scratch_bytes) return fail(q, PMR_EINVAL, "scratch too small", hipSuccess);
        HIPCHK(hipMemcpyAsync(q->d_scratch, b + src * elem, keep * elem, hipMemcpyDeviceToDevice, st), "shift");
        HIPCHK(hipMemcpyAsync(b, q->d_scratch, keep * elem, hipMemcpyDeviceToDevice, st), "shift");
    }
    return PMR_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* front end, staged: dc-block (:795) -> half-band cascade -> arbitrary resampler (:796)         */

int frontend_staged(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out)
{
    const pmr_design *d = &q->d;
    const unsigned h = d->num_stages;
    *ny_out = 0;
    if (n_in == 0) return PMR_OK;

    const unsigned ntiles = (n_in + PMR_DC_TILE - 1) / PMR_DC_TILE;
    const unsigned l_last = n_in - (ntiles - 1) * PMR_DC_TILE;
    const float lam_last = (float)pow(d->dc_lambda, (double)l_last);
    const float inv_last = (float)pow(d->dc_lambda, -(double)(PMR_DC_TILE - l_last));
    LAUNCH_FE(K_DC_AGG, pmr_launch_dc_agg(q->sfe, d_iq, n_in, q->d_dc_agg, &q->dcc, q->d_lam_thread_pow));
    LAUNCH_FE(K_DC_SCAN, pmr_launch_dc_scan(q->sfe, q->d_dc_agg, ntiles, q->d_dc_W, q->d_dc_state, &q->dcc,
                                         q->d_lam_tile_idx_pow, lam_last, inv_last));
    LAUNCH_FE(K_DC_APPLY, pmr_launch_dc_apply(q->sfe, d_iq, n_in, q->d_dc_W, q->d_z[0] + q->keep[0], &q->dcc,
                                           q->d_lam_thread_pow));

    uint64_t c_e = q->n_raw;         /* absolute count of z_e samples before this call */
    unsigned n_e = n_in;             /* new z_e samples this call                      */
    for (unsigned e = 0; e < h; e++) {
        const unsigned g = h - 1 - e;
        const unsigned n_out = (unsigned)(((c_e + n_e) >> 1) - (c_e >> 1));
        const int par = (int)(c_e & 1u);
        LAUNCH_FE(K_HALFBAND, pmr_launch_halfband(q->sfe, q->d_z[e], q->d_z[e + 1] + q->keep[e + 1], n_out,
                                               (int)q->keep[e], par, (int)d->m_stage[g], q->d_hb_h1[g],
                                               e == h - 1 ? d->zeta : 1.0f));
        int rc = shift_front(q, q->sfe, q->d_z[e], sizeof(cfl), n_e, q->keep[e]);
        if (rc) return rc;
        c_e >>= 1; n_e = n_out;
    }
    /* n_e new decimated samples in z_h; resamp_crcf phase bookkeeping */
    const uint64_t span = (uint64_t)n_e << 24;
    unsigned ny = 0;
    if (n_e && (uint64_t)q->arb_phase < span)
        ny = (unsigned)((span - q->arb_phase + d->arb_step - 1) / d->arb_step);
    LAUNCH_FE(K_ARB, pmr_launch_arb(q->sfe, q->d_z[h], q->d_xr, q->xr_abs, q->xr_mask, ny, q->arb_phase,
                                    d->arb_step, q->d_arb_bank, (int)q->keep[h]));
    q->arb_phase = (uint32_t)((uint64_t)q->arb_phase + (uint64_t)ny * d->arb_step - span);
    int rc = shift_front(q, q->sfe, q->d_z[h], sizeof(cfl), n_e, q->keep[h]);
    if (rc) return rc;
    *ny_out = ny;
    return PMR_OK;
}

/* branch taps of stages [e0, e0 + n) into the kernel-argument copy (specialised front-end kernel) */
static void fe_fill_taps(const struct pmr_chain_s *q, pmr_fe_params *p, unsigned e0, unsigned n)
{
    unsigned total = 0;
    for (unsigned e = e0; e < e0 + n; e++) total += 2u * (unsigned)q->fe_m[e];
    p->taps_valid = 0;
    if (n == 0 || total > sizeof(p->taps_k) / sizeof(p->taps_k[0])) return;
    memcpy(p->taps_k, q->fe_taps_host + q->fe_tap_off[e0], total * sizeof(float));
    p->taps_valid = 1;
}

/* carry bookkeeping shared by the fused and the two-level front end: parameters of the tile-carry sum (k_fe_tiles /
 * k_fe_tilefix / k_fe_carry) for a launch of `ntiles` tiles whose first tile starts `pend` samples before the block */
static void fe_carry_params(const struct pmr_chain_s *q, pmr_fe_tiles_params *t, unsigned slot, unsigned ntiles, unsigned c_end,
                            int off_end, unsigned pend, int cur, int nxt)
{
    const double lam = q->d.dc_lambda;
    memset(t, 0, sizeof(*t));
    t->probeA = q->d_fe_probeA + (size_t)slot * q->fe_max_tiles; t->probeB = q->d_fe_probeB + (size_t)slot * q->fe_max_tiles;
    t->probeL = q->d_fe_probeL + slot; t->probeE = q->d_fe_probeE + slot;
    t->v_in = q->d_fe_vstate[cur]; t->v_out = q->d_fe_vstate[nxt]; t->V = q->d_fe_V[slot];
    t->ntiles = ntiles; t->K = q->fe_K; t->c_end = c_end;
    t->rho = (float)pow(lam, (double)q->fe_T_own);
    t->lamHh = (float)pow(lam, (double)q->fe_Hh); t->inv_lamHh = (float)pow(lam, -(double)q->fe_Hh);
    t->inv_lamL = (float)pow(lam, -(double)(q->fe_Hh + (int)pend)); t->lamEnd = (float)pow(lam, (double)off_end + 1.0);
    t->rho_pow = q->d_fe_rho_pow;
    t->tile_j = q->d_fe_tile_j + (size_t)slot * 2 * q->fe_max_tiles;
}

/* floor(2^56 / step), clamped to 32 bits: the kernels' integer ceil-division by the resampler step (no fp64 on the device) */
static uint32_t step_rinv(uint32_t step)
{
    const uint64_t r = step ? (1ull << 56) / step : 0;
    return r > 0xffffffffull ? 0xffffffffu : (uint32_t)r;
}

/* front end, fused: one pass over the raw block (pmr_fe_fast.hip / pmr_frontend.hip) */
int frontend_fused(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out)
{
    const pmr_design *d = &q->d;
    const unsigned h = d->num_stages, D = d->decim;
    *ny_out = 0;
    if (n_in == 0) return PMR_OK;
    const unsigned pend = (unsigned)(q->n_raw & (D - 1));
    const unsigned Q = (unsigned)(((q->n_raw + n_in) >> h) - (q->n_raw >> h));
    unsigned ny = 0, ns_unused = 0; uint32_t new_phase = 0;
    plan_core(h, d->arb_step, q->M, q->n_raw, q->arb_phase, 0, n_in, &ny, &ns_unused, &new_phase);
    const unsigned long total = (unsigned long)pend + n_in;
    const unsigned ntiles = (unsigned)((total + q->fe_T_own - 1) / q->fe_T_own);
    const unsigned c_end = (unsigned)((total - 1) / q->fe_T_own);
    const int off_end = (int)((total - 1) - (unsigned long)c_end * q->fe_T_own) + q->fe_Hh;
    if (ntiles > q->fe_max_tiles) return fail(q, PMR_ERANGE, "tile count", hipSuccess);
    const int cur = q->fe_sel, nxt = cur ^ 1;
    const unsigned slot = (unsigned)(q->n_calls % PIPE_DEPTH);

    pmr_fe_tiles_params t;
    fe_carry_params(q, &t, slot, ntiles, c_end, off_end, pend, cur, nxt);
    pmr_fe_params p;
    memset(&p, 0, sizeof(p));
    p.x = d_iq; p.in_fmt = q->cur_in_fmt; p.lds_pad = q->fe_lds_pad;
    p.hist = q->d_fe_hist[cur]; p.new_hist = q->d_fe_hist[nxt]; p.out = q->d_xr; p.out_pos0 = q->xr_abs; p.out_mask = q->xr_mask;
    p.probeA = (void *)t.probeA; p.probeB = (void *)t.probeB; p.probeL = (void *)t.probeL; p.probeE = (void *)t.probeE;
    p.tile_j = (void *)t.tile_j;
    p.hb_taps = q->d_fe_taps; p.arb_bank = q->d_arb_bank; p.lam_lane_pow = q->d_fe_lam_lane;
    p.n_in = n_in; p.ny = ny; p.Q = Q; p.phi0 = q->arb_phase; p.step = d->arb_step; p.step_rinv = step_rinv(d->arb_step);
    p.h = (int)h; p.T_own = q->fe_T_own; p.Hh = q->fe_Hh; p.HhQ = q->fe_HhQ; p.TQ = q->fe_TQ;
    p.pend = (int)pend; p.hcap = q->fe_hcap; p.c_end = (int)c_end; p.off_end = off_end;
    memcpy(p.m, q->fe_m, sizeof(p.m)); memcpy(p.tap_off, q->fe_tap_off, sizeof(p.tap_off));
    p.dc_a1 = d->dc_a1; p.zeta = d->zeta; p.lam_wave = q->fe_lam_wave;
    memcpy(p.lam_pow16, q->fe_lam_pow16, sizeof(p.lam_pow16));
    fe_fill_taps(q, &p, 0, h);
    {
        pmr_launch_events ev; prof_pending pe;
        fe_launch_events(q, K_FE, q->tf_on_backend && ntiles != 0, &ev, &pe);
        LAUNCH_FE(K_FE, pmr_launch_frontend(q->sfe, &p, ntiles, q->fe_nt, q->fe_spt, &ev));
        prof_push(q, &pe);
    }

    pmr_fe_fix_params f;
    memset(&f, 0, sizeof(f));
    f.xr = q->d_xr; f.pos0 = q->xr_abs; f.mask = q->xr_mask; f.V = q->d_fe_V[slot]; f.GA = q->d_fe_GAK; f.T1 = q->d_fe_T1; f.T2 = q->d_fe_T2;
    f.ny = ny; f.TQ = (unsigned)q->fe_TQ; f.HhQ = (unsigned)q->fe_HhQ; f.phi0 = q->arb_phase; f.step = d->arb_step;
    f.Kgain = q->fe_Kgain;
    if (q->cal_now) {
        /* carry applied at the channelizer's loads: here only the tail later calls re-read as history is corrected in place
         * (every sample from (frames_done' - p) M on, frames_done' M >= end - (M - 1)) */
        const unsigned keep = (q->d.pfb_p + 3u) * q->M;
        f.j0 = ny > keep ? ny - keep : 0;
        q->cal_fix_limit = f.j0; q->cal_ntiles = ntiles; q->cal_slot = slot; q->cal_phi0 = q->arb_phase;
    }
    if (q->tf_on_backend) {
        /* the carry pass heads the back-end stream's work for this block; the front-end stream then
         * carries front-end kernels only, back to back */
        q->pend_t2 = t; q->pend_f2 = f; q->pend_tf_Q = Q; q->pend_tf = 1;
    } else {
        pmr_launch_events ev; prof_pending pe;
        fe_launch_events(q, K_FE_TILEFIX, t.ntiles != 0, &ev, &pe);
        if (q->cal_now) LAUNCH_FE(K_FE_TILEFIX, pmr_launch_fe_carry_tail(q->sfe, &t, &f, &ev));
        else LAUNCH_FE(K_FE_TILEFIX, pmr_launch_fe_tilefix(q->sfe, &t, &f, Q, &ev));
    }
    q->fe_sel = nxt;
    q->arb_phase = new_phase;
    *ny_out = ny;
    return PMR_OK;
}

/* front end, two levels (deep cascades): level 1 = dc-block + first s1 (six-tap) stages -> d_fe_ring1 on the FRONT-END
 * stream; then, on the back-end stream when there is one (`defer`), k_fe_carry (tile carries of level 1 + in-place dc fix of
 * the ring tail the next call re-reads as history) and level 2 = remaining stages + resampler reading that ring with the
 * carry applied at load (k_fe_level2, or the generic k_frontend in mode 2). */
int frontend_two_level(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out)
{
    const pmr_design *d = &q->d;
    const unsigned h = d->num_stages, s1 = (unsigned)q->fe_s1, h2 = h - s1, D1 = 1u << s1, D2 = 1u << h2;
    *ny_out = 0;
    if (n_in == 0) return PMR_OK;
    const unsigned pend1 = (unsigned)(q->n_raw & (D1 - 1));
    const uint64_t A = q->n_raw >> s1;                                   /* level-1 output samples before this call */
    const unsigned Q1 = (unsigned)(((q->n_raw + n_in) >> s1) - A);       /* ... produced by this call               */
    const unsigned Q = (unsigned)(((q->n_raw + n_in) >> h) - (q->n_raw >> h));
    unsigned ny = 0, ns_unused = 0; uint32_t new_phase = 0;
    plan_core(h, d->arb_step, q->M, q->n_raw, q->arb_phase, 0, n_in, &ny, &ns_unused, &new_phase);
    const int cur = q->fe_sel, nxt = cur ^ 1;
    const unsigned slot = (unsigned)(q->n_calls % PIPE_DEPTH);

    /* ---- level 1 ---- */
    const unsigned long total1 = (unsigned long)pend1 + n_in;
    const unsigned ntiles1 = (unsigned)((total1 + q->fe_T_own - 1) / q->fe_T_own);
    const unsigned c_end = (unsigned)((total1 - 1) / q->fe_T_own);
    const int off_end = (int)((total1 - 1) - (unsigned long)c_end * q->fe_T_own) + q->fe_Hh;
    if (ntiles1 > q->fe_max_tiles) return fail(q, PMR_ERANGE, "tile count", hipSuccess);
    pmr_fe_tiles_params t;
    fe_carry_params(q, &t, slot, ntiles1, c_end, off_end, pend1, cur, nxt);
    pmr_fe_params p;
    memset(&p, 0, sizeof(p));
    p.mode = 1;
    p.x = d_iq; p.in_fmt = q->cur_in_fmt; p.hist = q->d_fe_hist[cur]; p.new_hist = q->d_fe_hist[nxt];
    p.out = q->d_fe_ring1; p.out_pos0 = A; p.out_mask = q->ring1_mask;
    p.probeA = (void *)t.probeA; p.probeB = (void *)t.probeB; p.probeL = (void *)t.probeL; p.probeE = (void *)t.probeE;
    p.hb_taps = q->d_fe_taps; p.arb_bank = q->d_arb_bank; p.lam_lane_pow = q->d_fe_lam_lane;
    p.n_in = n_in; p.ny = 0; p.Q = Q1; p.phi0 = 0; p.step = 1;
    p.h = (int)s1; p.T_own = q->fe_T_own; p.Hh = q->fe_Hh; p.HhQ = q->fe_HhQ; p.TQ = q->fe_TQ;
    p.pend = (int)pend1; p.hcap = q->fe_hcap; p.c_end = (int)c_end; p.off_end = off_end;
    memcpy(p.m, q->fe_m, sizeof(p.m)); memcpy(p.tap_off, q->fe_tap_off, sizeof(p.tap_off));
    p.dc_a1 = d->dc_a1; p.zeta = 1.0f; p.lam_wave = q->fe_lam_wave;
    memcpy(p.lam_pow16, q->fe_lam_pow16, sizeof(p.lam_pow16));
    fe_fill_taps(q, &p, 0, s1);
    /* (level 1 is launched below, once the parameters of the block's carry pass and level 2 are made: host arithmetic only) */

    /* ---- carries of level 1 + in-place fix of the ring tail: the last `keep` new samples are what the NEXT call's level 2
     * re-reads as history; level 2 of THIS call skips them (fix_limit) and corrects everything before them at load ---- */
    const unsigned keep = (unsigned)q->fe2_Hh + D2 + 16;
    pmr_fe_fix_params f;
    memset(&f, 0, sizeof(f));
    f.xr = q->d_fe_ring1; f.pos0 = A; f.mask = q->ring1_mask; f.V = q->d_fe_V[slot]; f.GA = q->d_fe_GA;
    f.T1 = q->d_fe_T1; f.T2 = q->d_fe_T2; f.ny = Q1; f.j0 = Q1 > keep ? Q1 - keep : 0;
    f.TQ = (unsigned)q->fe_TQ; f.HhQ = (unsigned)q->fe_HhQ; f.phi0 = 0; f.step = 0; f.Kgain = q->fe1_K;

    /* ---- level 2: Q1 new samples of the decimated ring -> last h2 stages -> resampler ---- */
    const unsigned pend2 = (unsigned)(A & (D2 - 1));
    const unsigned long total2 = (unsigned long)pend2 + Q1;
    const unsigned ntiles2 = Q1 ? (unsigned)((total2 + q->fe2_T_own - 1) / q->fe2_T_own) : 0;
    pmr_fe_params p2;
    memset(&p2, 0, sizeof(p2));
    p2.mode = 2;
    p2.in_ring = q->d_fe_ring1; p2.in_mask = q->ring1_mask; p2.in_abs0 = (int64_t)A;
    p2.fixV = q->d_fe_V[slot]; p2.fix_T1 = q->d_fe_T1; p2.fix_T2 = q->d_fe_T2; p2.fix_G = q->d_fe_G1;
    p2.fix_rTQ = 1.0f / (float)q->fe_TQ;
    p2.fix_TQ = (unsigned)q->fe_TQ; p2.fix_HhQ = (unsigned)q->fe_HhQ; p2.fix_K = q->fe1_K; p2.fix_limit = f.j0;
    p2.out = q->d_xr; p2.out_pos0 = q->xr_abs; p2.out_mask = q->xr_mask;
    p2.hb_taps = q->d_fe_taps; p2.arb_bank = q->d_arb_bank; p2.lam_lane_pow = q->d_fe_lam_lane;
    p2.n_in = Q1; p2.ny = ny; p2.Q = Q; p2.phi0 = q->arb_phase; p2.step = d->arb_step; p2.step_rinv = step_rinv(d->arb_step);
    p2.h = (int)h2; p2.T_own = q->fe2_T_own; p2.Hh = q->fe2_Hh; p2.HhQ = q->fe2_HhQ; p2.TQ = q->fe2_TQ;
    p2.pend = (int)pend2; p2.hcap = 0; p2.c_end = (int)ntiles2 - 1; p2.off_end = 0;
    for (unsigned e = 0; e < h2; e++) { p2.m[e] = q->fe_m[s1 + e]; p2.tap_off[e] = q->fe_tap_off[s1 + e]; }
    p2.dc_a1 = d->dc_a1; p2.zeta = d->zeta; p2.lam_wave = q->fe_lam_wave;
    memcpy(p2.lam_pow16, q->fe_lam_pow16, sizeof(p2.lam_pow16));
    fe_fill_taps(q, &p2, s1, h2);
#ifdef EXP_L2_INLINE     /* timing experiment (pmr_fe_fast.hip): level-2 tiles run inside the level-1 launch; the separate launch keeps the
                          * tiles a real implementation could not place there (the first LAG level-1 tiles of every XCD range).  WRONG results */
    {
        extern void pmr_exp_set_l2_params(const pmr_fe_params *);
        pmr_exp_set_l2_params(&p2);
    }
#endif
    {
        /* with level 2 deferred to the back-end stream, level 1 is the front-end stream's last launch of this call */
        pmr_launch_events ev; prof_pending pe;
        fe_launch_events(q, K_FE, q->l2_on_backend && ntiles1 != 0, &ev, &pe);
        LAUNCH_FE(K_FE, pmr_launch_frontend(q->sfe, &p, ntiles1, 256, 16, &ev));
        prof_push(q, &pe);
    }
#ifdef EXP_L2_INLINE
    unsigned ntiles2_sep = ntiles2;
    {
        const unsigned per = ntiles1 / 8, lag = EXP_L2_INLINE;
        const unsigned long inl = per > lag ? (unsigned long)(per - lag) * 8ul * (unsigned)q->fe_TQ / (unsigned)q->fe2_T_own : 0;
        ntiles2_sep = inl < ntiles2 ? ntiles2 - (unsigned)inl : 0;
    }
#define ntiles2 ntiles2_sep
#endif
    if (q->l2_on_backend) {
        /* Level 2 touches 1/2^s1 of the data in a few thousand tiles -- too few to fill the chip -- so it runs best under the
         * next block's level 1 instead of between two level-1 launches on the same stream. */
        q->pend_t2 = t; q->pend_f2 = f; q->pend_p2 = p2; q->pend_ntiles2 = ntiles2; q->pend_l2 = 1;
    } else {
        LAUNCH_FE(K_FE_TILES, pmr_launch_fe_carry(q->sfe, &t, &f));
        if (ntiles2) LAUNCH_FE(K_FE_L2, pmr_launch_frontend_l2(q->sfe, &p2, ntiles2, q->fe2_fast));
    }
#ifdef EXP_L2_INLINE
#undef ntiles2
#endif
    q->fe_sel = nxt;
    q->arb_phase = new_phase;
    *ny_out = ny;
    return PMR_OK;
}

/* front end only, for pmr_dsd.c (pmr_internal.h): everything on stream_fe, dc carry applied in place */
int pmr_chain_frontend_block(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out, uint64_t *xr_abs0)
{
    if (!q) return PMR_EINVAL;
    if (n_in > q->cfg.max_block) return fail(q, PMR_ERANGE, "n_in > max_block", hipSuccess);
    if (n_in && !d_iq) return fail(q, PMR_EINVAL, "null input", hipSuccess);
    unsigned ny_plan = 0, ns_plan = 0, ny = 0;
    plan_counts(q, n_in, &ny_plan, &ns_plan);
    if (ny_plan > q->res_size) return fail(q, PMR_ERANGE, "resampled stream overflow", hipSuccess);
    *xr_abs0 = q->xr_abs;
    const int keep_l2 = q->l2_on_backend, keep_tf = q->tf_on_backend;
    q->l2_on_backend = 0;                         /* this entry point has no back-end stream: everything on stream_fe */
    q->tf_on_backend = 0;
    q->cal_now = 0;                               /* ... and no channelizer: the carry is applied in place */
    q->sfe = q->stream_fe;
    q->fe_done_ev = NULL; q->fe_done_used = 0;
    int rc = !q->fe_on ? frontend_staged(q, d_iq, n_in, &ny)
                       : q->fe_two ? frontend_two_level(q, d_iq, n_in, &ny) : frontend_fused(q, d_iq, n_in, &ny);
    q->l2_on_backend = keep_l2; q->tf_on_backend = keep_tf;
    if (rc) return rc;
    if (ny != ny_plan) return fail(q, PMR_EINVAL, "internal: resampler count mismatch", hipSuccess);
    q->n_raw += n_in;
    q->xr_abs += ny;
    q->frames_done = q->xr_abs / q->M;
    q->last_ny = ny;
    q->n_calls++;
    *ny_out = ny;
    return PMR_OK;
}

unsigned pmr_chain_plan_resampled(pmr_chain q, unsigned n_in)
{
    unsigned ny = 0, ns = 0;
    plan_counts(q, n_in, &ny, &ns);
    return ny;
}

void pmr_chain_frontend_view(pmr_chain q, pmr_fe_view *v)
{
    v->d_xr = q->d_xr; v->xr_mask = q->xr_mask; v->stream_fe = (void *)q->stream_fe; v->d_in = q->d_in;
    v->res_size = q->res_size; v->device = q->device;
}
