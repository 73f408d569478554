/* pmr_chain.c -- host side of libpmr446_hip.so: the process-one-block entry points of include/pmr_chain.h.
 *
 * Plain C (the reference's host language).  Owns the per-stream device state that the liquid objects of
 * reference include/sdr_pmr446.h:54-82 (struct _proc_chain_t) own on the CPU, keeps the closed-form sample
 * counters on the host (so no device->host sync is ever needed to size a launch), and enqueues the gfx950
 * kernels of pmr_kernels.hip on one HIP stream.  There is NO CPU fallback: without a HIP device
 * pmr_chain_create() fails.
 */
#include "pmr_chain_priv.h"


/* ------------------------------------------------------------------------------------------- */

int fail(pmr_chain q, int code, const char *what, hipError_t e)
{
    if (q) snprintf(q->err, sizeof(q->err), "%s%s%s", what, e != hipSuccess ? ": " : "",
                    e != hipSuccess ? hipGetErrorString(e) : "");
    /* ANY failure in the middle of a block (q->in_block: between the first launch / counter update of a block and its last) leaves
     * the stream position undefined -- the host counters may be ahead of what the device did, whether the cause was the HIP runtime,
     * an allocation or a capacity check that could only be made mid-way: the handle refuses further blocks until pmr_chain_reset */
    if (q && q->in_block) q->faulted = 1;
    return code;
}

int refuse_faulted(pmr_chain q)
{
    return fail(q, PMR_EHIP, "an earlier block failed mid-way: the stream position is undefined, call pmr_chain_reset", hipSuccess);
}

/* Two kinds of device buffer.  STATE (filter histories, rings whose older indices are "the samples before the stream began",
 * carried sums): zero is part of the algorithm -- dev_alloc_state.  SCRATCH (everything a kernel writes before another reads it):
 * zero-filled too, so that a run is reproducible, EXCEPT in the test-only poison mode (pmr_debug_poison, pmr_poison.hip), where
 * scratch is filled with 0xFF bytes (NaN as f32, -1 as an integer): a kernel that reads scratch nobody wrote then fails loudly
 * instead of passing on zeros. */
static int dev_alloc_fill(pmr_chain q, void **p, size_t bytes, int fill)
{
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(q, PMR_ENOMEM, "hipMalloc", e);
    e = hipMemsetAsync(*p, fill, bytes, q->stream);
    if (e != hipSuccess) return fail(q, PMR_EHIP, "hipMemsetAsync", e);
    return PMR_OK;
}
int dev_alloc(pmr_chain q, void **p, size_t bytes) { return dev_alloc_fill(q, p, bytes, pmr_debug_poison_enabled() ? 0xFF : 0); }
int dev_alloc_state(pmr_chain q, void **p, size_t bytes) { return dev_alloc_fill(q, p, bytes, 0); }

int dev_upload(pmr_chain q, float **p, const float *src, size_t n)
{
    int rc = dev_alloc(q, (void **)p, n * sizeof(float));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(*p, src, n * sizeof(float), hipMemcpyHostToDevice, q->stream), "upload");
    HIPCHK(hipStreamSynchronize(q->stream), "upload sync");   /* src may be a temporary */
    return PMR_OK;
}

/* FIR tap table for the audio kernels: h zero-padded by PMR_TAP_PAD on both sides, natural order:
 * Q[PMR_TAP_PAD + d] = h[d].  Input step e (frame t0-J-(n-1)+e) meets accumulator i (frame t0-J+i) with
 * h[(n-1)+i-e] = Q[PMR_TAP_PAD + (n-1) + i - e]. */
int upload_padded_taps(pmr_chain q, float **p, const float *h, unsigned n)
{
    size_t len = n + 2 * PMR_TAP_PAD;
    float *tmp = (float *)calloc(len, sizeof(float));
    if (!tmp) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
    for (unsigned j = 0; j < n; j++) tmp[j + PMR_TAP_PAD] = h[j];
    int rc = dev_upload(q, p, tmp, len);
    free(tmp);
    return rc;
}

/* The environment switches of DESIGN.md 7a, read ONCE per handle, here; nothing on a launch path calls getenv.  Round 4 cut them
 * down to what a user or a test has a reason to flip: the exact direct form of the audio FIR (the FFT form's reference), single-stream
 * calls, the copy-engine path of small synchronous calls, and the in-place form of the dc carry (what the at-load form must equal
 * bit for bit).  Kernel variants that lost their A/B live in git history and in tools/variant_bench.sh builds, not in the product. */
static int env_is(const char *name, const char *val) { const char *e = getenv(name); return e && !strcmp(e, val); }
static void read_switches(pmr_switches *w)
{
    memset(w, 0, sizeof(*w));
    w->fir_direct = env_is("PMR_FIR", "direct");
    w->fir_fft4096 = env_is("PMR_FIR", "fft4096");
    w->fir_fft2048 = env_is("PMR_FIR", "fft2048");
    w->fir_fft1024 = env_is("PMR_FIR", "fft1024");
    w->no_overlap = env_is("PMR_OVERLAP", "0");
    w->carry_inplace = env_is("PMR_CARRY", "inplace");
    w->no_zerocopy = env_is("PMR_ZEROCOPY", "0");
}

static pmr_chain chain_create(const pmr_chain_cfg *cfg, int frontend_only);
pmr_chain pmr_chain_create(const pmr_chain_cfg *cfg) { return chain_create(cfg, 0); }
pmr_chain pmr_chain_create_frontend(const pmr_chain_cfg *cfg) { return chain_create(cfg, 1); }

static __thread char g_create_err[256];
static void create_error(const char *what) { snprintf(g_create_err, sizeof(g_create_err), "%s", what); }

static pmr_chain chain_create(const pmr_chain_cfg *cfg, int frontend_only)
{
    g_create_err[0] = 0;
    if (!cfg) { create_error("null configuration"); return NULL; }
    if (cfg->num_channels < 2 && !frontend_only) {
        fprintf(stderr, "pmr_chain_create: invalid configuration\n");
        create_error("invalid configuration: num_channels < 2");
        return NULL;
    }
    pmr_chain q = (pmr_chain)calloc(1, sizeof(*q));
    if (!q) { create_error("out of memory"); return NULL; }
    q->cfg = *cfg;
    read_switches(&q->sw);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "pmr_chain_create: no HIP device (this library has no CPU path)\n");
        create_error("no HIP device (this library has no CPU path)");
        free(q); return NULL;
    }
    if (cfg->device >= 0) {
        if (cfg->device >= ndev) {
            snprintf(g_create_err, sizeof(g_create_err), "device ordinal %d does not exist (%d HIP device%s visible)", cfg->device, ndev, ndev == 1 ? "" : "s");
            free(q); return NULL;
        }
        if (hipSetDevice(cfg->device) != hipSuccess) { create_error("hipSetDevice failed"); free(q); return NULL; }
        q->device = cfg->device;
    } else if (cfg->device != -1) {
        snprintf(g_create_err, sizeof(g_create_err), "device ordinal %d: -1 (the calling thread's current device) or 0 .. %d", cfg->device, ndev - 1);
        free(q); return NULL;
    } else if (hipGetDevice(&q->device) != hipSuccess) { create_error("hipGetDevice failed"); free(q); return NULL; }
    if (cfg->max_block == 0 ||
        pmr_design_build(&q->d, cfg->fs_in, cfg->num_channels, cfg->channel_width_hz, cfg->dcblock_alpha,
                         cfg->resamp_As, cfg->pfb_m, cfg->pfb_As, cfg->fm_kf)) {
        fprintf(stderr, "pmr_chain_create: invalid configuration\n");
        create_error("invalid configuration");
        pmr_design_free(&q->d); free(q); return NULL;
    }
    q->M = cfg->num_channels;
    pmr_design_buffer_sizes(&q->d, cfg->max_block, &q->res_size, &q->chan_size);
    /* Stream priorities (A/B history: profiles/r03_stream_priority.txt, r03_ab_log.txt r43-r46, r04_ab_log.txt r4d / r4e).  Equal for
     * one-level plans: the back-end stream is critical at cfg2 (front end high: -9 %), cfg3 is indifferent.  Two-level plans put the
     * FRONT-END stream high (cfg5 +1.5-2 % on three boxes) -- but only when the audio FIR of large blocks is the FFT form, whose
     * one-wave, 8.7 KB workgroups fit beside four level-1 tiles: round 3's direct form (36 KB of LDS per workgroup) starved behind a
     * high-priority front end (541 -> 495 GS/s in steady state, bimodal regions).  Which form runs is only known once chain_init has
     * made the plan (fe_two really selected, fft_ok: not PMR_FIR=direct, not deemph_fir / lowpass, <= 512 taps), so the front-end
     * stream is created AFTER it; chain_init itself queues on q->stream only. */
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);       /* numerically lower = higher priority */
    /* The base priority is NORMAL (0), not the range's least (1 on ROCm 7.2, what rounds 1-2 used for both streams): a process that
     * had held a handle with a high-priority stream and then created a handle with two LEAST-priority streams saw those two
     * serialise (cfg2 276 instead of 381 GS/s as bench.py's second workload) -- they apparently end up on one hardware queue.  With
     * normal / high that does not happen (profiles/r03_stream_priority.txt). */
    const int prio_base = (prio_hi <= 0 && 0 <= prio_lo) ? 0 : prio_lo;
    if (hipStreamCreateWithPriority(&q->stream, hipStreamNonBlocking, prio_base) != hipSuccess) {
        create_error("hipStreamCreateWithPriority failed");
        pmr_design_free(&q->d); free(q); return NULL;
    }
    if (hipEventCreateWithFlags(&q->ev_switch, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&q->stream_h2d, hipStreamNonBlocking) != hipSuccess) { create_error("stream / event creation failed"); pmr_chain_destroy(q); return NULL; }
    if (hipStreamCreateWithPriority(&q->stream_ct, hipStreamNonBlocking, prio_base) != hipSuccess ||
        hipEventCreateWithFlags(&q->ev_ctlp, hipEventDisableTiming) != hipSuccess) { create_error("stream / event creation failed"); pmr_chain_destroy(q); return NULL; }
    for (unsigned i = 0; i < PIPE_DEPTH; i++) {
        if (hipEventCreateWithFlags(&q->ev_fe[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q->ev_ct[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q->ev_be[i], hipEventDisableTiming) != hipSuccess) {
            create_error("event creation failed");
            pmr_chain_destroy(q); return NULL;
        }
    }
    q->overlap = !q->sw.no_overlap;
    if (chain_init(q) != PMR_OK) {
        fprintf(stderr, "pmr_chain_create: %s\n", q->err);
        create_error(q->err);
        pmr_chain_destroy(q);
        return NULL;
    }
    /* (round 6: also the one-level plan whose tiles are PADDED to leave the back end's workgroup its LDS -- the 256-channel plan,
     * fe_lds_pad: the front end cannot starve a back end whose room is reserved; cfg3 +1.2 %, 3 of 3 interleaved pairs, while the
     * unpadded 16-channel plan loses 10 % with a high-priority front end: profiles/r06_ab_log.txt r6h) */
    q->fe_prio_high = q->fe_on && q->fft_ok && (q->fe_two || q->fe_lds_pad != 0);
#ifdef EXP_FE_PRIO_EQUAL   /* A/B hooks: both streams at the base priority in every plan / the front-end stream high in every plan */
    q->fe_prio_high = 0;
#endif
#ifdef EXP_FE_PRIO_HIGH
    q->fe_prio_high = q->fe_on;
#endif
    if (hipStreamCreateWithPriority(&q->stream_fe, hipStreamNonBlocking, q->fe_prio_high ? prio_hi : prio_base) != hipSuccess) {
        create_error("hipStreamCreateWithPriority failed");
        pmr_chain_destroy(q);
        return NULL;
    }
    q->sfe = q->stream_fe;
    return q;
}

/* Why the last pmr_chain_create / pmr_chain_create_frontend of THIS thread returned NULL ("" after a success): the handle that would
 * carry pmr_chain_last_error does not exist then. */
const char *pmr_chain_create_error(void) { return g_create_err; }

int pmr_chain_destroy(pmr_chain q)
{
    if (!q) return PMR_OK;
    hipSetDevice(q->device);
    if (q->stream_fe) hipStreamSynchronize(q->stream_fe);
    if (q->stream) hipStreamSynchronize(q->stream);
    prof_resolve(q);
    for (unsigned i = 0; i < PIPE_DEPTH; i++) { if (q->ev_fe[i]) hipEventDestroy(q->ev_fe[i]); if (q->ev_be[i]) hipEventDestroy(q->ev_be[i]); if (q->ev_ct[i]) hipEventDestroy(q->ev_ct[i]); }
    if (q->ev_ctlp) hipEventDestroy(q->ev_ctlp);
    if (q->stream_ct) { hipStreamSynchronize(q->stream_ct); hipStreamDestroy(q->stream_ct); }
    if (q->ev_switch) hipEventDestroy(q->ev_switch);
    if (q->stream_h2d) { hipStreamSynchronize(q->stream_h2d); hipStreamDestroy(q->stream_h2d); }
    if (q->stream_fe) hipStreamDestroy(q->stream_fe);
    for (unsigned i = 0; i < q->npool; i++) hipEventDestroy(q->pool[i]);
    free(q->pool); free(q->pend);
    for (unsigned g = 0; g < PMR_MAX_STAGES; g++) if (q->d_hb_h1[g]) hipFree(q->d_hb_h1[g]);
    for (unsigned e = 0; e <= PMR_MAX_STAGES; e++) if (q->d_z[e]) hipFree(q->d_z[e]);
    void *bufs[] = { q->d_arb_bank, q->d_pfb_taps_t, q->d_fft_tw, q->d_nco_cs, q->d_lam_thread_pow,
                     q->d_lam_tile_idx_pow, q->d_hp_pad, q->d_lp_pad, q->d_de_pad, q->d_in, q->d_dc_state,
                     q->d_dc_agg, q->d_dc_W, q->d_xr, q->d_fm, q->d_aux1, q->d_aux2, q->d_scratch,
                     q->d_chan_x, q->d_chan_list, q->d_reset_flags, q->d_rssi_part, q->d_dbg_xr, q->d_dbg_fm, q->d_dbg_ct, q->d_fe_taps, q->d_fe_GA,
                     q->d_fe_T1, q->d_fe_T2, q->d_fe_lam_lane, q->d_fe_hist[0], q->d_fe_hist[1], q->d_fe_vstate[0],
                     q->d_fe_vstate[1], q->d_fe_probeA, q->d_fe_probeB, q->d_fe_probeL, q->d_fe_probeE, q->d_fe_V[0],
                     q->d_fe_V[1], q->d_fe_V[2], q->d_fe_G12, q->d_fe_GAK, q->d_fe_ring1, q->d_fe_tile_j, q->d_fe_rho_pow, q->d_ctlp, q->d_ct_taps, q->d_ct_taps_ext, q->d_ct_lampow, q->d_ct_agg, q->d_ct_W, q->d_ct_dcstate,
                     q->d_ct_U, q->d_ct_coef, q->d_ct_part, q->d_ct_carry[0], q->d_ct_carry[1], q->d_ct_events, q->d_ct_restart,
                     q->d_spec_win, q->d_spec_tw, q->d_spec_part, q->d_spec_psd, q->d_fe_G1,
                     q->d_fft_H[0], q->d_fft_H[1], q->d_fft_H2[0], q->d_fft_H2[1], q->d_fft_TA[0], q->d_fft_TA[1], q->d_fft_TB[0], q->d_fft_TB[1],
                     q->d_fft_H[2], q->d_fft_H2[2], q->d_fft_TA[2], q->d_fft_TB[2] };
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); i++) if (bufs[i]) hipFree(bufs[i]);
    for (unsigned i = 0; i < PIPE_DEPTH; i++) {
        pmr_slot *sl = &q->slot[i];
        void *dv[] = { i ? (void *)sl->d_in : NULL, sl->d_out, sl->d_chan };
        void *hv[] = { sl->h_out, sl->h_chan };
        for (size_t j = 0; j < sizeof(dv) / sizeof(dv[0]); j++) if (dv[j]) hipFree(dv[j]);
        for (size_t j = 0; j < sizeof(hv) / sizeof(hv[0]); j++) if (hv[j]) hipHostFree(hv[j]);
        if (sl->d_raw) hipFree(sl->d_raw);
        if (sl->done) hipEventDestroy(sl->done);
        if (sl->in_ready) hipEventDestroy(sl->in_ready);
    }
    if (q->stream) hipStreamDestroy(q->stream);
    pmr_design_free(&q->d);
    free(q->h_reset_flags);
    free(q->h_open);
    free(q->ct_open_last);
    for (unsigned i = 0; i < PIPE_DEPTH; i++) free(q->slot[i].open_rows);
    free(q);
    return PMR_OK;
}

int pmr_chain_reset(pmr_chain q)
{
    if (!q) return PMR_EINVAL;
    const unsigned M = q->M, h = q->d.num_stages;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    /* in-flight work of un-synchronised device calls (front end on stream_fe, back end on stream) finishes first:
     * nothing may overwrite the zeroed state afterwards */
    HIPCHK(hipStreamSynchronize(q->stream_fe), "reset");
    HIPCHK(hipStreamSynchronize(q->stream), "reset");
    HIPCHK(hipStreamSynchronize(q->stream_ct), "reset");
    for (unsigned i = 0; i < PIPE_DEPTH; i++) q->ct_ev_used[i] = 0;
    q->ct_async_last = 0;
    HIPCHK(hipMemsetAsync(q->d_dc_state, 0, sizeof(cfl), q->stream), "reset");
    for (unsigned e = 0; e <= h; e++)
        HIPCHK(hipMemsetAsync(q->d_z[e], 0, (size_t)q->keep[e] * sizeof(cfl), q->stream), "reset");
    HIPCHK(hipMemsetAsync(q->d_xr, 0, (size_t)(q->xr_mask + 1) * sizeof(cfl), q->stream), "reset");
    HIPCHK(hipMemsetAsync(q->d_fm, 0, (size_t)(q->fm_mask + 1) * M * sizeof(float), q->stream), "reset");
    if (q->d_aux1) {
        HIPCHK(hipMemsetAsync(q->d_aux1, 0, (size_t)(q->fm_mask + 1) * M * sizeof(float), q->stream), "reset");
        HIPCHK(hipMemsetAsync(q->d_aux2, 0, (size_t)(q->fm_mask + 1) * M * sizeof(float), q->stream), "reset");
    }
    if (q->d_ctlp) {
        HIPCHK(hipMemsetAsync(q->d_ctlp, 0, (size_t)(q->fm_mask + 1) * M * sizeof(float), q->stream), "reset");
        HIPCHK(hipMemsetAsync(q->d_ct_dcstate, 0, (size_t)M * sizeof(float), q->stream), "reset");
        for (int i = 0; i < 2; i++)
            HIPCHK(hipMemsetAsync(q->d_ct_carry[i], 0, (size_t)M * PMR_CT_TONES * 2 * sizeof(float), q->stream), "reset");
        HIPCHK(hipMemsetAsync(q->d_ct_restart, 0, M, q->stream), "reset");
    }
    if (q->d_fe_ring1) HIPCHK(hipMemsetAsync(q->d_fe_ring1, 0, (size_t)(q->ring1_mask + 1) * sizeof(cfl), q->stream), "reset");
    if (q->fe_on) for (int i = 0; i < 2; i++) {
        HIPCHK(hipMemsetAsync(q->d_fe_hist[i], 0, (size_t)q->fe_hcap * sizeof(cfl), q->stream), "reset");
        HIPCHK(hipMemsetAsync(q->d_fe_vstate[i], 0, sizeof(cfl), q->stream), "reset");
    }
    q->fe_sel = 0;
    q->n_raw = 0; q->arb_phase = 0; q->xr_abs = 0; q->frames_done = 0; q->n_calls = 0; q->last_ny = q->last_ns = 0;
    q->pend_l2 = 0; q->pend_tf = 0; q->tf_last_be = 0; q->pend_audio = 0;
    q->reset_pending = 0; memset(q->h_reset_flags, 0, M);
    HIPCHK(hipStreamSynchronize(q->stream_h2d), "reset");
    q->slot_head = 0; q->n_inflight = 0;                   /* blocks submitted but not collected are dropped */
    q->faulted = 0; q->in_block = 0;
    for (unsigned i = 0; i < PIPE_DEPTH; i++) q->slot[i].used = 0;
    HIPCHK(hipStreamSynchronize(q->stream), "reset sync");
    return PMR_OK;
}

/* Stream position n_raw with every filter state zero -- exactly the state after n_raw ZERO samples (a linear chain fed zeros stays
 * at zero; the discriminator's arg(0) = 0): reset, then the closed-form counters of plan_core taken over the whole prefix in one
 * step.  Everything else that depends on the position is derived from them when a block is planned: the front end's pending raw
 * samples (n_raw mod 2^h), the resampler phase, the ring positions (absolute indices), the NCO phase (xr index mod 2M), the frame
 * remainder (xr_abs - frames_done M) and the detector's 2441-frame grid (frames_done).  The reference's loop never ends
 * (src/sdr_pmr446.c:788): at cfg5 n_raw passes 2^32 after 4.3 s -- this is how tests put a handle there without streaming to it. */
int pmr_chain_seek(pmr_chain q, uint64_t n_raw)
{
    if (!q) return PMR_EINVAL;
    if (n_raw >> 62) return fail(q, PMR_ERANGE, "seek position", hipSuccess);
    int rc = pmr_chain_reset(q);
    if (rc) return rc;
    const unsigned __int128 span = (unsigned __int128)(n_raw >> q->d.num_stages) << 24;
    const uint64_t ny = (uint64_t)((span + q->d.arb_step - 1u) / q->d.arb_step);      /* phase 0 at the origin: plan_core */
    q->n_raw = n_raw;
    q->arb_phase = (uint32_t)((unsigned __int128)ny * q->d.arb_step - span);
    q->xr_abs = ny;
    q->frames_done = q->M ? ny / q->M : 0;
    return PMR_OK;
}

void pmr_chain_position(pmr_chain q, uint64_t *n_raw, uint64_t *n_resampled, uint64_t *n_frames)
{
    if (n_raw) *n_raw = q ? q->n_raw : 0;
    if (n_resampled) *n_resampled = q ? q->xr_abs : 0;
    if (n_frames) *n_frames = q ? q->frames_done : 0;
}

unsigned pmr_chain_max_frames(pmr_chain q) { return q ? q->chan_size : 0; }
unsigned pmr_chain_num_channels(pmr_chain q) { return q ? q->M : 0; }
const char *pmr_chain_last_error(pmr_chain q) { return q ? q->err : "null handle"; }
void *pmr_chain_stream(pmr_chain q) { return q ? (void *)q->stream : NULL; }

/* d_iq of pmr_chain_process_block_device is READ by the front end, which pipelined calls queue on a second stream: a caller
 * that produces d_iq on a stream of its own hands over an event recorded behind its producer; the next call's reads wait for it. */
int pmr_chain_wait_input_event(pmr_chain q, void *hip_event)
{
    if (!q || !hip_event) return PMR_EINVAL;
    q->input_ready = (hipEvent_t)hip_event;
    q->has_input_ready = 1;
    return PMR_OK;
}

/* returns when every queued block's front end has finished reading its d_iq: the buffers may be overwritten */
int pmr_chain_synchronize_input(pmr_chain q)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    HIPCHK(hipStreamSynchronize(q->stream_fe), "hipStreamSynchronize");
    if (q->last_single) HIPCHK(hipStreamSynchronize(q->stream), "hipStreamSynchronize");
    return PMR_OK;
}

int pmr_chain_synchronize(pmr_chain q)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipStreamSynchronize(q->stream_fe), "hipStreamSynchronize");
    HIPCHK(hipStreamSynchronize(q->stream), "hipStreamSynchronize");
    HIPCHK(hipStreamSynchronize(q->stream_ct), "hipStreamSynchronize");
    prof_resolve(q);
    return PMR_OK;
}

/* copy `n` elements starting at absolute ring index `pos` into a linear device buffer (debug capture) */
int ring_to_linear(pmr_chain q, void *dst, const void *ring, uint64_t mask, uint64_t pos, size_t n, size_t elem)
{
    const uint64_t cap = mask + 1, i0 = pos & mask;
    const size_t first = (size_t)((cap - i0) < n ? (cap - i0) : n);
    HIPCHK(hipMemcpyAsync(dst, (const char *)ring + i0 * elem, first * elem, hipMemcpyDeviceToDevice, q->stream), "dbg");
    if (first < n)
        HIPCHK(hipMemcpyAsync((char *)dst + first * elem, ring, (n - first) * elem, hipMemcpyDeviceToDevice, q->stream), "dbg");
    return PMR_OK;
}

/* Audio part of one block (frames frame0 .. frame0 + ns of the discriminator ring): HP (:882) -> gain (:890) -> de-emphasis
 * (:895-899) -> optional LP (:900-902) -> sink (:903-906), the CTCSS branch when the detector is on -- for the channels the
 * mask has open NOW. */
int audio_part(pmr_chain q, int64_t frame0, unsigned ns, void *d_pcm, void *d_audio, unsigned pcm_stride)
{
    const unsigned M = q->M;
    int rc;
    /* audio: HP (:882) -> gain (:890) -> de-emphasis (:895-899) -> optional LP (:900-902) -> sink (:903-906); with the CTCSS
     * detector on, its low-pass branch delay188(x) - hp(x) (:884-889) is a second tap set over the same samples: one pass */
    int ct_fir_done = 0, rssi_rode = 0, audio_done = 0;
    {
        /* large blocks: overlap-save FFT form (pmr_fir_fft.hip); with the detector on its low-pass branch is the second product */
        const int dual = q->ct_on && q->fft_ok && q->fft_tab[0].H2 != NULL;
        const int which = (d_pcm || d_audio) && (!q->ct_on || dual) ? fir_fft_pick(q, ns, q->mask_on ? q->n_enabled : M, dual) : -1;
        if (which >= 0) {
            LAUNCH(K_FIR_HP, pmr_launch_fir_fft(q->stream, which, &q->fft_tab[which], q->d_fm, q->fm_mask, frame0, ns, M, q->hp_len,
                                                (int16_t *)d_pcm, (float *)d_audio, pcm_stride, dual ? q->d_ctlp : NULL,
                                                q->mask_on ? q->d_chan_list : NULL, q->n_enabled));
            audio_done = 1; ct_fir_done = dual;
        }
    }
    if (!audio_done && q->ct_on && q->d_ct_taps_ext && (d_pcm || d_audio) && !q->cfg.deemph_fir && !q->cfg.lowpass) {
        prof_pending pp_; prof_begin(q, K_FIR_HP, &pp_, q->stream);
        const int rd = pmr_launch_fir_dual(q->stream, q->d_fm, q->fm_mask, frame0, ns, M, q->d_hp_pad, q->d_ct_taps_ext,
                                           q->hp_len, (int16_t *)d_pcm, (float *)d_audio, pcm_stride, q->d_ctlp,
                                           q->mask_on ? q->d_chan_list : NULL, q->n_enabled);
        prof_end(q, &pp_, q->stream);
        if (rd > 0) return fail(q, PMR_EHIP, pmr_k_names[K_FIR_HP], (hipError_t)rd);
        ct_fir_done = rd == 0;
    }
    if (q->ct_on && (rc = ctcss_run(q, frame0, ns, ct_fir_done))) return rc;

    if (!audio_done && !ct_fir_done && (d_pcm || d_audio || q->cfg.deemph_fir || q->cfg.lowpass)) {
        const int more = q->cfg.deemph_fir || q->cfg.lowpass;
        /* only the open channels are demodulated to audio.  With follow-on FIR passes (deemph_fir / lowpass) the mask
         * applies to the LAST pass only: the intermediate rings must keep every channel's history current, or a channel
         * opened later would start from a cold filter */
        const unsigned *sel = q->mask_on ? q->d_chan_list : NULL;
        LAUNCH(K_FIR_HP, pmr_launch_fir_tm(q->stream, q->d_fm, q->fm_mask, frame0, ns, M, q->d_hp_pad, q->hp_len,
                                           1.0f, 0, 0.f, 0.f, 0.f,      /* gain + de-emphasis are in the taps */
                                           more ? q->d_aux1 : NULL, more ? NULL : (int16_t *)d_pcm,
                                           more ? NULL : (float *)d_audio, pcm_stride, more ? NULL : sel, q->n_enabled,
                                           q->rssi_job_pending ? &q->rssi_job : NULL, &rssi_rode));
        const float *cur = q->d_aux1;
        const int sink = d_pcm || d_audio;                 /* (none: a pending block is only pushed through the stateful passes) */
        if (q->cfg.deemph_fir && (sink || q->cfg.lowpass)) {
            const int last = !q->cfg.lowpass;
            LAUNCH(K_FIR_DE, pmr_launch_fir_tm(q->stream, cur, q->fm_mask, frame0, ns, M, q->d_de_pad, q->de_len,
                                               1.0f, 0, 0.f, 0.f, 0.f, last ? NULL : q->d_aux2,
                                               last ? (int16_t *)d_pcm : NULL, last ? (float *)d_audio : NULL,
                                               pcm_stride, last ? sel : NULL, q->n_enabled, NULL, NULL));
            cur = q->d_aux2;
        }
        if (q->cfg.lowpass && sink) {
            LAUNCH(K_FIR_LP, pmr_launch_fir_tm(q->stream, cur, q->fm_mask, frame0, ns, M, q->d_lp_pad, q->lp_len,
                                               1.0f, 0, 0.f, 0.f, 0.f, NULL, (int16_t *)d_pcm, (float *)d_audio,
                                               pcm_stride, sel, q->n_enabled, NULL, NULL));
        }
    }
    if (q->rssi_job_pending) {
        q->rssi_job_pending = 0;
        if (!rssi_rode)
            LAUNCH(K_RSSI, pmr_launch_rssi_finish(q->stream, q->rssi_job.rssi_part, q->rssi_job.ntiles, M, q->rssi_job.ns, q->rssi_job.rssi_db));
    }
    return PMR_OK;
}

/* Which stream carries the one-level form's carry pass (k_fe_tilefix, 0.022 ms + a kernel boundary)?  It only needs the block's
 * front end before it and the channelizer after it, so it can close the front-end stream's work for the block or open the
 * back-end stream's.  The two streams are balanced within a few per cent, so it belongs on the lighter one.  Front-end load per
 * input sample is constant; back-end load grows with the resampled rate r = M * 12.5 kHz / fs_in and with the share of channels
 * that are demodulated to audio (the FIR is ~3/4 of it; twice the work with the CTCSS detector on).  Measured on MI355X
 * (tools/env_ab2.sh, 2^26-sample blocks):   cfg3 all channels (r = 0.052): 356 -> 378 GS/s on the back-end stream;
 * cfg2 all channels (r = 0.083): 352 -> 317;   one open channel: cfg2 404 -> 422, cfg3 408 -> 415.
 * (Round 4, cfg2 with the carry at load: on the front-end stream 442 / 424 vs 422 / 424 GS/s -- bimodal, not taken.) */
static int tilefix_on_backend(const pmr_chain q)
{
    const double r = (double)q->M * q->cfg.channel_width_hz / q->cfg.fs_in;
    const double f_open = q->mask_on ? (double)q->n_enabled / (double)q->M : 1.0;
    const double load = r * (1.0 + 3.0 * f_open) * (q->ct_on ? 2.0 : 1.0);
    return load < 0.25;
}

/* `single`: queue the whole block on ONE stream (no cross-stream events): what a caller that synchronises after every
 * block wants -- the two-stream pipeline only pays when consecutive blocks are in flight together. */
static int process_block_device_body(pmr_chain q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio,
                                     unsigned pcm_stride, unsigned *n_frames, void *d_chan_out, void *d_rssi_db, int single, int phase);

/* every exit path of a block leaves in_block clear; an error return with in_block set has already marked the handle faulted (fail) */
int process_block_device_impl(pmr_chain q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio,
                                     unsigned pcm_stride, unsigned *n_frames, void *d_chan_out, void *d_rssi_db, int single,
                                     int phase /*0: whole block; 1: up to channelizer + RSSI, audio part left pending*/)
{
    if (!q) return PMR_EINVAL;
    if (q->faulted) return refuse_faulted(q);
    const int rc = process_block_device_body(q, d_iq, n_in, d_pcm, d_audio, pcm_stride, n_frames, d_chan_out, d_rssi_db, single, phase);
    if (rc && q->in_block) q->faulted = 1;
    q->in_block = 0;
    return rc;
}

static int process_block_device_body(pmr_chain q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio,
                                     unsigned pcm_stride, unsigned *n_frames, void *d_chan_out, void *d_rssi_db, int single, int phase)
{
    q->cur_single = single;
    q->rssi_job_pending = 0;
    if (phase == 1 && !single) return fail(q, PMR_EINVAL, "two-step form is synchronous", hipSuccess);
    if (q->pend_audio) {
        /* a channelized block was never demodulated: filters that carry state of their own through the audio part (CTCSS
         * detector, follow-on FIR passes) must still see it */
        q->pend_audio = 0;
        if (q->ct_on || q->cfg.deemph_fir || q->cfg.lowpass) {
            q->in_block = 1;
            int rc_ = audio_part(q, q->pend_audio_frame0, q->pend_audio_ns, NULL, NULL, 0);
            if (rc_) return rc_;
            q->in_block = 0;
        }
    }
    if (n_in > q->cfg.max_block) return fail(q, PMR_ERANGE, "n_in > max_block", hipSuccess);
    if (n_in && !d_iq) return fail(q, PMR_EINVAL, "null input", hipSuccess);
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    const pmr_design *d = &q->d;
    const unsigned M = q->M, p = d->pfb_p;
    int rc;

    /* validate against the closed-form counts BEFORE any state is advanced */
    unsigned ny_plan = 0, ns_plan = 0;
    plan_counts(q, n_in, &ny_plan, &ns_plan);
    if (n_frames) *n_frames = ns_plan;
    if (ns_plan > q->chan_size) return fail(q, PMR_ERANGE, "frame count exceeds max_frames", hipSuccess);
    if (ns_plan > pcm_stride && (d_pcm || d_audio || d_chan_out)) return fail(q, PMR_ERANGE, "stride < frames", hipSuccess);
    if (ny_plan > q->res_size) return fail(q, PMR_ERANGE, "resampled stream overflow", hipSuccess);
    q->in_block = 1;                             /* from here on a runtime failure poisons the handle (fail()) */

    /* ---- front end of this block on stream_fe.  It may run while the back end of the PREVIOUS block is still
     * busy on q->stream; it must not start before the back end of the block before that has released its part
     * of the rings (they hold history + PIPE_DEPTH blocks). ---- */
    const unsigned par = (unsigned)(q->n_calls % PIPE_DEPTH);
    q->cur_par = par;
    q->sfe = single ? q->stream : q->stream_fe;
    if (!single) {
        if (q->last_single && q->n_calls) {      /* everything the single-stream calls queued on q->stream comes first */
            HIPCHK(hipEventRecord(q->ev_switch, q->stream), "record");
            HIPCHK(hipStreamWaitEvent(q->stream_fe, q->ev_switch, 0), "wait single-stream calls");
        }
        if (q->n_calls >= PIPE_DEPTH) {
            /* ring reuse: the back end of block n - PIPE_DEPTH must be done.  The HOST waits for it (back-pressure: at most
             * PIPE_DEPTH blocks are ever queued) -- a wait packet on the front-end stream instead (round 2's PMR_HOST_GATE=0 form) sits between
             * two front-end launches and costs 3 % at cfg5 (439 vs 454 GS/s, tools/env_ab.sh) */
            HIPCHK(hipEventSynchronize(q->ev_be[par]), "wait back end");
        }
        if (q->ct_ev_used[par]) {                /* ... and its CTCSS detector (own stream): it reads the low-pass ring's rows */
            q->ct_ev_used[par] = 0;
            HIPCHK(hipEventSynchronize(q->ev_ct[par]), "wait detector");
        }
    }
    if (single && q->ct_async_last)              /* pipelined calls' detectors (own stream) still read the rings this call writes */
        HIPCHK(hipStreamWaitEvent(q->stream, q->ev_ct[q->ct_last_par], 0), "wait detector");
    q->last_single = single;
    if (q->has_input_ready) {                    /* the caller's producer of d_iq finishes first (pmr_chain_wait_input_event) */
        q->has_input_ready = 0;
        HIPCHK(hipStreamWaitEvent(q->sfe, q->input_ready, 0), "wait input event");
    }
    const uint64_t xr_abs0 = q->xr_abs;
    unsigned ny = 0;
    /* the ring must hold corrected samples when something besides the channelizer reads it (debug capture, waterfall) */
    q->cal_now = q->cal_ok && !q->dbg_on && !q->spec_nfft;
    q->cal_fix_limit = 0;
    if (q->fe_on && !q->fe_two) {
        q->tf_on_backend = !single && (q->cal_now || tilefix_on_backend(q));
        if (!single && !q->tf_on_backend && q->tf_last_be)    /* this block's carry pass reads the dc state the previous one (back-end stream) wrote */
            HIPCHK(hipStreamWaitEvent(q->stream_fe, q->ev_be[(par + PIPE_DEPTH - 1) % PIPE_DEPTH], 0), "wait previous carry pass");
        q->tf_last_be = q->tf_on_backend;
    }
    q->fe_done_ev = single ? NULL : q->ev_fe[par];
    q->fe_done_used = 0;
    if ((rc = !q->fe_on ? frontend_staged(q, d_iq, n_in, &ny)
                        : q->fe_two ? frontend_two_level(q, d_iq, n_in, &ny) : frontend_fused(q, d_iq, n_in, &ny))) return rc;
    if (ny != ny_plan) return fail(q, PMR_EINVAL, "internal: resampler count mismatch", hipSuccess);
    /* "front end of this block finished": the completion signal of the front-end stream's last launch where that launch could
     * carry it, a record packet otherwise (staged kernels, level 2 on the front-end stream, empty blocks, profiled launches) */
    if (!single && !q->fe_done_used) HIPCHK(hipEventRecord(q->ev_fe[par], q->stream_fe), "record");
    q->n_raw += n_in;
    q->xr_abs += ny;
    q->last_ny = ny;

    /* ---- back end on q->stream ---- */
    if (!single) HIPCHK(hipStreamWaitEvent(q->stream, q->ev_fe[par], 0), "wait front end");
    if (q->pend_tf) {
        q->pend_tf = 0;
#ifdef EXP_NO_TAIL      /* timing experiment (tools/ab_libs.py): what would the chain gain without this launch?  WRONG results */
        if (q->cal_now) { }
        else
#endif
        if (q->cal_now) LAUNCH(K_FE_TILEFIX, pmr_launch_fe_carry_tail(q->stream, &q->pend_t2, &q->pend_f2, NULL));
        else LAUNCH(K_FE_TILEFIX, pmr_launch_fe_tilefix(q->stream, &q->pend_t2, &q->pend_f2, q->pend_tf_Q, NULL));
    }
    if (q->pend_l2) {
        q->pend_l2 = 0;
#ifndef EXP_NO_CARRY5   /* timing experiment: the two-level plan without its carry launch.  WRONG results */
        LAUNCH(K_FE_TILES, pmr_launch_fe_carry(q->stream, &q->pend_t2, &q->pend_f2));
#endif
#ifndef EXP_SKIP_L2
        if (q->pend_ntiles2) LAUNCH(K_FE_L2, pmr_launch_frontend_l2(q->stream, &q->pend_p2, q->pend_ntiles2, q->fe2_fast));
#endif
    }
    if (q->dbg_on && ny)
        if ((rc = ring_to_linear(q, q->d_dbg_xr, q->d_xr, q->xr_mask, xr_abs0, ny, sizeof(cfl)))) return rc;
    if (q->spec_nfft) {                           /* asgramcf_write(resamp_buf, ny) + execute (:911-912): PSD of THIS block's samples */
        q->spec_ntr_last = ny / (q->spec_nfft / 2);
        LAUNCH(K_SPGRAM, pmr_launch_spgram(q->stream, q->d_xr, q->xr_mask, xr_abs0, ny, q->spec_nfft, q->d_spec_win, q->d_spec_tw,
                                           q->d_spec_part, q->d_spec_psd));
    }

    /* ring carry (:797,:804): frames of M samples, 0..M-1 remainder stays for the next call */
    const unsigned ns = (unsigned)((q->xr_abs - q->frames_done * M) / M);
    const int64_t frame0 = (int64_t)q->frames_done;
    q->last_ns = ns;
    q->ct_nev_last = 0;

    if (ns) {
        unsigned ntiles = 0;
        pmr_chan_params c;
        memset(&c, 0, sizeof(c));
        c.xr = q->d_xr; c.xr_mask = q->xr_mask; c.frame0 = frame0; c.xr_end = q->xr_abs;
        c.fm = q->d_fm; c.fm_mask = q->fm_mask; c.ns = ns; c.M = M; c.p = p;
        c.taps_t = q->d_pfb_taps_t; c.fft_tw = q->d_fft_tw; c.nco_cs = q->d_nco_cs; c.nco_period = d->nco_period;
        c.fm_ref = d->fm_ref; c.chan_out = d_chan_out; c.chan_stride = pcm_stride;
        c.rssi_part = d_rssi_db ? q->d_rssi_part : NULL;
        if (q->cal_now && q->cal_fix_limit) {
            c.fix.V = q->d_fe_V[q->cal_slot]; c.fix.GA = q->d_fe_GAK; c.fix.G12 = q->d_fe_G12;
            c.fix.pos0 = xr_abs0; c.fix.phi0 = q->cal_phi0; c.fix.step = d->arb_step; c.fix.fix_limit = q->cal_fix_limit;
            c.fix.ntiles = q->cal_ntiles; c.fix.TQ = (unsigned)q->fe_TQ; c.fix.HhQ = (unsigned)q->fe_HhQ;
            c.fix.nbias = q->cal_nbias; c.fix.qbias = q->cal_nbias * (unsigned)q->fe_TQ; c.fix.nv = q->cal_nv;
        }
        if (q->reset_pending) {                       /* freqdem_reset (:866) of the flagged channels: takes effect on this call's first frame */
            HIPCHK(hipMemcpyAsync(q->d_reset_flags, q->h_reset_flags, M, hipMemcpyHostToDevice, q->stream), "reset flags");
            c.reset_flags = q->d_reset_flags;
        }
#ifndef EXP_SKIP_CHAN   /* timing experiments (tools/ab_libs.py): the chain without its channelizer / audio FIR launch.  WRONG results */
        if (q->chan_small) LAUNCH(K_CHANNELIZE_SMALL, pmr_launch_channelize_small(q->stream, &c, &ntiles));
        else if (q->chan_wide) LAUNCH(K_CHANNELIZE, pmr_launch_channelize_wide(q->stream, &c, q->d_chan_x, &ntiles));
        else LAUNCH(K_CHANNELIZE, pmr_launch_channelize(q->stream, &c, &ntiles));
#endif
        if (q->reset_pending) { q->reset_pending = 0; memset(q->h_reset_flags, 0, M); }
        if (q->dbg_on) {
            /* discriminator rows of this block, time-major, linearised */
            if ((rc = ring_to_linear(q, q->d_dbg_fm, q->d_fm, q->fm_mask, (uint64_t)frame0, ns, (size_t)M * sizeof(float))))
                return rc;
        }
        if (d_rssi_db) {
            if (phase != 1 && !q->dbg_on) {
                /* the RSSI finish rides in the audio FIR's launch where that kernel takes it (blocks of a few tiles: audio_part) */
                q->rssi_job.rssi_part = q->d_rssi_part; q->rssi_job.ntiles = ntiles; q->rssi_job.M = M; q->rssi_job.ns = ns;
                q->rssi_job.rssi_db = (float *)d_rssi_db; q->rssi_job_pending = 1;
            } else
                LAUNCH(K_RSSI, pmr_launch_rssi_finish(q->stream, q->d_rssi_part, ntiles, M, ns, (float *)d_rssi_db));
        }

#ifndef EXP_SKIP_FIR
        if (phase != 1 && (rc = audio_part(q, frame0, ns, d_pcm, d_audio, pcm_stride))) return rc;
#endif
    }
    if (phase == 1) { q->pend_audio = 1; q->pend_audio_frame0 = frame0; q->pend_audio_ns = ns; }
    q->frames_done += ns;
    if (!single) HIPCHK(hipEventRecord(q->ev_be[par], q->stream), "record");
    q->n_calls++;
    q->in_block = 0;
    return PMR_OK;
}

int pmr_chain_process_block_device(pmr_chain q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio,
                                   unsigned pcm_stride, unsigned *n_frames, void *d_chan_out, void *d_rssi_db)
{
    return process_block_device_impl(q, d_iq, n_in, d_pcm, d_audio, pcm_stride, n_frames, d_chan_out, d_rssi_db,
                                     q ? !q->overlap : 0, 0);
}

/* ------------------------------------------------------------------------------------------- */

/* ctcss_detector_reset of one channel (:348-357 via :867): u0 = u1 = 0 for every tone, samp_processed = 0.  The partial sums of
 * the block in progress restart from zero; the 2441-frame block GRID stays the stream's (every channel shares it here, where the
 * reference's single detector restarts its own count), so the block in progress is incomplete for this channel: flagged, and
 * k_ct_final reports its event as "no decision" {-1, 0, 0, 0}.  Known deviation: the reference's first event after a reset comes
 * exactly 2441 frames later, ours at the first grid boundary at least 2441 frames later (up to 2 x 2441 - 1).  Callers have
 * synchronised the back-end and detector streams. */
static hipError_t ct_restart_channel(pmr_chain q, unsigned k)
{
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; b++)
        e = hipMemset((char *)q->d_ct_carry[b] + (size_t)k * PMR_CT_TONES * 2 * sizeof(float), 0, (size_t)PMR_CT_TONES * 2 * sizeof(float));
    /* "restarted in mid-block" is relative to where the detector runs NEXT: in the two-step form (channelize -> mask / reset ->
     * demodulate) frames_done has already advanced past the pending block, whose audio part starts at pend_audio_frame0 */
    const uint64_t next_frame = q->pend_audio ? (uint64_t)q->pend_audio_frame0 : q->frames_done;
    if (e == hipSuccess) e = hipMemset(q->d_ct_restart + k, next_frame % PMR_CT_BLOCK ? 1 : 0, 1);
    return e;
}

/* Open-channel mask: the reference demodulates only the squelch-selected channel (src/sdr_pmr446.c:876-877, hand-off from the
 * squelch state machine :834-839).  Channelizer, RSSI and the discriminator keep running for EVERY channel (their state and the
 * audio filters' history therefore stay current for a channel that is opened later); the audio FIR / PCM / CTCSS branch run for
 * the enabled channels only. */
int pmr_chain_set_channel_mask(pmr_chain q, const uint64_t *mask_words, unsigned n_words)
{
    if (!q) return PMR_EINVAL;
    const unsigned M = q->M;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (!mask_words) {
        /* every channel opens: the detector of each channel that was closed restarts, like the explicit-list path below */
        if (q->mask_on && q->d_ct_carry[0]) {
            HIPCHK(hipStreamSynchronize(q->stream), "channel mask");
            HIPCHK(hipStreamSynchronize(q->stream_ct), "channel mask");
            for (unsigned k = 0; k < M; k++)
                if (!q->h_open[k]) HIPCHK(ct_restart_channel(q, k), "channel mask");
        }
        q->mask_on = 0; q->n_enabled = M; memset(q->h_open, 1, M);
        return PMR_OK;
    }
    if ((uint64_t)n_words * 64 < M) return fail(q, PMR_EINVAL, "channel mask shorter than num_channels", hipSuccess);
    /* the mask is applied by the MFMA audio kernels (16 channels per tile); the VALU versions (num_channels not a multiple of 16,
     * PMR_FIR=pair|lds|global) would write every row: refuse instead of breaking the "closed rows stay untouched" promise */
    if (!pmr_fir_mfma4_supported(M, q->hp_len))
        return fail(q, PMR_EINVAL, "channel mask needs the MFMA audio kernels (num_channels a multiple of 16)", hipSuccess);
    unsigned *list = (unsigned *)malloc((size_t)M * sizeof(unsigned));
    if (!list) return fail(q, PMR_ENOMEM, "malloc", hipSuccess);
    unsigned n = 0;
    for (unsigned k = 0; k < M; k++) if (mask_words[k >> 6] >> (k & 63) & 1ull) list[n++] = k;
    /* the list is read by kernels of calls already queued: let them finish before it changes */
    hipError_t e = hipStreamSynchronize(q->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(q->stream_ct);          /* the detector reads the list too */
    if (e == hipSuccess && n) e = hipMemcpy(q->d_chan_list, list, (size_t)n * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess && q->d_ct_carry[0]) {
        /* ctcss_detector_reset of the channels that open now (:867): their partial Goertzel sums were frozen while closed */
        for (unsigned i = 0; i < n && e == hipSuccess; i++) {
            if (q->h_open[list[i]] && q->mask_on) continue;                /* was open already */
            if (!q->mask_on) break;                                        /* every channel was running */
            e = ct_restart_channel(q, list[i]);
        }
    }
    if (e == hipSuccess) {
        memset(q->h_open, 0, M);
        for (unsigned i = 0; i < n; i++) q->h_open[list[i]] = 1;
    }
    free(list);
    if (e != hipSuccess) return fail(q, PMR_EHIP, "channel mask upload", e);
    q->n_enabled = n;
    q->mask_on = n < M;
    return PMR_OK;
}

/* freqdem_reset + ctcss_detector_reset of ONE channel (the reference resets its single demodulator when the squelch detunes,
 * src/sdr_pmr446.c:866-867).  The discriminator's previous sample becomes zero, so the channel's first output of the next call
 * is arg(0) = 0; the channel's partial Goertzel sums restart from zero (the 2441-frame block grid itself stays
 * aligned to the stream: every channel shares it here).  Takes effect on the next process_block call that yields a frame. */
int pmr_chain_reset_channel(pmr_chain q, unsigned channel)
{
    if (!q || channel >= q->M) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    q->h_reset_flags[channel] = 1;
    q->reset_pending = 1;
    if (q->d_ct_carry[0]) {
        HIPCHK(hipStreamSynchronize(q->stream), "reset channel");
        HIPCHK(hipStreamSynchronize(q->stream_ct), "reset channel");
        HIPCHK(ct_restart_channel(q, channel), "reset channel");
        /* (the detector's dc blocker, :606, is NOT reset: the reference's detune path calls freqdem_reset and ctcss_detector_reset
         * only, :866-867 -- ctcss_dcblock keeps its state) */
    }
    return PMR_OK;
}

int pmr_chain_set_overlap(pmr_chain q, int on)
{
    if (!q) return PMR_EINVAL;
    int rc = pmr_chain_synchronize(q);
    q->overlap = on ? 1 : 0;
    return rc;
}

unsigned pmr_chain_info(pmr_chain q, int what, unsigned idx)
{
    if (what == PMR_INFO_EXPERIMENT_BUILD) return (unsigned)(PMR_EXPERIMENT_BUILD | pmr_kernels_experiment_build());
    if (!q) return 0;
    switch (what) {
    case PMR_INFO_NUM_STAGES: return q->d.num_stages;
    case PMR_INFO_M_STAGE:    return idx < q->d.num_stages ? q->d.m_stage[idx] : 0;
    case PMR_INFO_ARB_STEP:   return q->d.arb_step;
    case PMR_INFO_NCO_DTHETA: return q->d.nco_dtheta;
    case PMR_INFO_ARB_NPFB:   return PMR_ARB_NPFB;
    case PMR_INFO_ARB_M:      return PMR_ARB_M;
    case PMR_INFO_PFB_P:      return q->d.pfb_p;
    case PMR_INFO_CARRY_AT_LOAD: return (unsigned)q->cal_ok;
    case PMR_INFO_FE_PLAN:
        if (!q->fe_on) return 0;
        if (!q->fe_two) return q->fe_fast_fmt ? 2 : 1;
        return q->fe_fast_fmt && q->fe2_fast ? 3 : 4;
    case PMR_INFO_CHAN_PLAN: return q->chan_small ? 1 : q->chan_wide ? (q->M == 256 ? 2 : 3) : 0;
    case PMR_INFO_FIR_PLAN: return !pmr_fir_mfma4_supported(q->M, q->hp_len) ? 0 : q->fft_ok ? 2 : 1;
    default: return 0;
    }
}

unsigned pmr_chain_design(pmr_chain q, int what, unsigned idx, float *out, unsigned cap)
{
    if (!q) return 0;
    const float *src = NULL; unsigned n = 0;
    switch (what) {
    case PMR_DESIGN_HALFBAND:
        if (idx >= q->d.num_stages) return 0;
        src = q->d.hb_proto[idx]; n = 4 * q->d.m_stage[idx] + 1; break;
    case PMR_DESIGN_ARB: src = q->d.arb_proto; n = 2 * PMR_ARB_M * PMR_ARB_NPFB + 1; break;
    case PMR_DESIGN_PFB: src = q->d.pfb_proto; n = 2 * q->M * q->d.pfb_m + 1; break;
    case PMR_DESIGN_DEEMPH: {
        const float de[3] = { q->d.de_b0, q->d.de_b1, q->d.de_a1 };
        if (out) memcpy(out, de, (size_t)(cap < 3 ? cap : 3) * sizeof(float));
        return 3;
    }
    default: return 0;
    }
    if (out) memcpy(out, src, (size_t)(n < cap ? n : cap) * sizeof(float));
    return n;
}
