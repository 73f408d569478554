/* pmr_chain_host.c -- the host-buffer entry points (the reference's call pattern, one readStream block per call: src/sdr_pmr446.c:789-796):
 * slots, synchronous / asynchronous / two-step calls, pinned memory. */
#include "pmr_chain_priv.h"


/* ---- host-buffer entry points ---------------------------------------------------------------------------------
 * A SLOT is one block in flight between host buffers: its own device input staging, device outputs and pinned host outputs.
 * The synchronous pmr_chain_process_block* use slot 0 on one stream (no cross-stream events: nothing overlaps anyway);
 * pmr_chain_submit_block / pmr_chain_collect_block cycle through PIPE_DEPTH slots so that the H2D copy, the kernels and the
 * D2H copy of consecutive blocks overlap (the call pattern of the reference's loop, one readStream block per iteration,
 * src/sdr_pmr446.c:789-796, with the sink one block behind).  Device outputs of a slot are COMPACT -- [M][stride] with
 * stride = frames of this block rounded up to 8 -- so the D2H copy is one contiguous transfer whatever M is (a 2-D copy of
 * 1024 rows of 100 bytes runs at a few hundred MB/s); the rows are then spread into the caller's [M][pcm_stride] layout by
 * the CPU. */
/* Outputs of a slot live in ONE device block and ONE pinned host block, laid out per call as [rssi | pcm | audio] (each part
 * 256-byte aligned, compact stride), so whatever subset was asked for comes back in a single D2H copy. */
#define SLOT_ALIGN(x) (((x) + 255u) & ~(size_t)255u)
static int slot_prepare(pmr_chain q, unsigned i, int want_chan)
{
    pmr_slot *sl = &q->slot[i];
    const size_t out_n = (size_t)q->M * ((q->chan_size + 7u) & ~7u);
    int rc;
    if (!sl->d_in) {
        if (i == 0) sl->d_in = q->d_in;
        else if ((rc = dev_alloc(q, (void **)&sl->d_in, (size_t)q->cfg.max_block * sizeof(cfl)))) return rc;
        sl->out_bytes = SLOT_ALIGN((size_t)q->M * sizeof(float)) + SLOT_ALIGN(out_n * sizeof(int16_t)) + SLOT_ALIGN(out_n * sizeof(float));
        if ((rc = dev_alloc(q, (void **)&sl->d_out, sl->out_bytes))) return rc;
        if (hipHostMalloc((void **)&sl->h_out, sl->out_bytes, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&sl->done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sl->in_ready, hipEventDisableTiming) != hipSuccess)
            return fail(q, PMR_ENOMEM, "pinned slot buffers", hipSuccess);
        if (hipHostGetDevicePointer((void **)&sl->hd_out, sl->h_out, 0) != hipSuccess) { sl->hd_out = NULL; (void)hipGetLastError(); }
        HIPCHK(hipStreamSynchronize(q->stream), "slot init");
    }
    if (want_chan && !sl->d_chan) {
        if ((rc = dev_alloc(q, (void **)&sl->d_chan, out_n * sizeof(cfl)))) return rc;
        if (hipHostMalloc((void **)&sl->h_chan, out_n * sizeof(cfl), hipHostMallocDefault) != hipSuccess)
            return fail(q, PMR_ENOMEM, "pinned slot buffers", hipSuccess);
        if (hipHostGetDevicePointer((void **)&sl->hd_chan, sl->h_chan, 0) != hipSuccess) { sl->hd_chan = NULL; (void)hipGetLastError(); }
        HIPCHK(hipStreamSynchronize(q->stream), "slot init");
    }
    return PMR_OK;
}

static const void *host_zero_copy(const void *p, size_t bytes);

/* remember which channels the audio part of the slot's block runs for: the FIR leaves the rows of closed channels alone, and the
 * compact staging rows they would come from hold another block's data */
static int slot_snapshot_mask(pmr_chain q, pmr_slot *sl)
{
    sl->masked = q->mask_on;
    if (!q->mask_on) return PMR_OK;
    if (!sl->open_rows && !(sl->open_rows = (uint8_t *)malloc(q->M))) return fail(q, PMR_ENOMEM, "malloc", hipSuccess);
    memcpy(sl->open_rows, q->h_open, q->M);
    return PMR_OK;
}

/* queue one block: H2D -> chain -> D2H into the slot's pinned buffers; nothing is waited for.
 * Synchronous calls on SMALL blocks skip both copy engines (each copy is a submission of its own with ~10 us of hand-over on
 * either side, 100 us -> 70 us per 100 000-sample call): the front end reads the caller's pinned buffer in place and the last
 * kernels write the slot's pinned output buffer directly (PMR_ZEROCOPY=0 restores the copies). */
static int slot_submit(pmr_chain q, unsigned i, const void *iq, int fmt, unsigned n_in, unsigned want, int single, int phase)
{
    pmr_slot *sl = &q->slot[i];
    int rc = slot_prepare(q, i, (want & PMR_WANT_CHAN) != 0);
    if (rc) return rc;
    if (n_in > q->cfg.max_block) return fail(q, PMR_ERANGE, "n_in > max_block", hipSuccess);
    if (n_in && !iq) return fail(q, PMR_EINVAL, "null input", hipSuccess);
    if (fmt < 0 || fmt > 2) return fail(q, PMR_EINVAL, "unknown IQ format", hipSuccess);
    if (fmt && !sl->d_raw && (rc = dev_alloc(q, &sl->d_raw, (size_t)q->cfg.max_block * 4))) return rc;
    unsigned ny_plan = 0, ns_plan = 0;
    plan_counts(q, n_in, &ny_plan, &ns_plan);
    const unsigned stride = ns_plan ? (ns_plan + 7u) & ~7u : 8u;
    const size_t n = (size_t)q->M * stride;
    sl->off_pcm = SLOT_ALIGN((size_t)q->M * sizeof(float));
    sl->off_audio = sl->off_pcm + SLOT_ALIGN(n * sizeof(int16_t));
    /* input: H2D (+ int16 / uint8 -> cf32 on the device).  Pipelined calls copy on their own stream, so the copy of block b+1
     * runs under the kernels of block b; it may not overwrite the slot's staging before the front end that last read it is done */
    hipStream_t s_in = single ? q->stream : q->stream_h2d;
    const cfl *d_iq = sl->d_in;
    int in_fmt = 0;                               /* format the front end is handed: != 0 only on the zero-copy path below */
    if (single && n_in && n_in <= ZC_MAX_IN && !q->sw.no_zerocopy && (fmt == 0 || q->fe_fast_fmt)) {
        /* the front end reads the caller's pinned buffer in place -- cf32, or the receiver's own int16 / uint8 samples converted as
         * the tile is loaded: 2 or 4 instead of 8 bytes per sample cross the host link, no copy-engine hand-over, no conversion pass */
        const void *z = host_zero_copy(iq, (size_t)n_in * (fmt == 0 ? 8 : fmt == 1 ? 4 : 2));
        if (z) { d_iq = (const cfl *)z; in_fmt = fmt; }
    }
    if (n_in && d_iq == sl->d_in) {
        if (!single && sl->used) HIPCHK(hipStreamWaitEvent(s_in, q->ev_fe[sl->par], 0), "wait front end");
        const size_t bytes = (size_t)n_in * (fmt == 0 ? 8 : fmt == 1 ? 4 : 2);
        HIPCHK(hipMemcpyAsync(fmt ? sl->d_raw : (void *)sl->d_in, iq, bytes, hipMemcpyHostToDevice, s_in), "H2D");
        if (fmt && (rc = pmr_launch_iq_convert(s_in, sl->d_raw, sl->d_in, n_in, fmt))) return fail(q, PMR_EHIP, "k_iq_convert", (hipError_t)rc);
        if (!single) {
            HIPCHK(hipEventRecord(sl->in_ready, s_in), "record");
            HIPCHK(hipStreamWaitEvent(q->stream_fe, sl->in_ready, 0), "wait input");
        }
    }
    sl->used = !single; sl->par = (unsigned)(q->n_calls % PIPE_DEPTH);
    unsigned ns = 0;
    const size_t out_hi = (want & PMR_WANT_AUDIO) ? sl->off_audio + n * sizeof(float) : sl->off_pcm + n * sizeof(int16_t);
    const int zc_out = single && !q->sw.no_zerocopy && sl->hd_out && out_hi <= ZC_MAX_OUT &&
                       (!(want & PMR_WANT_CHAN) || (sl->hd_chan && n * sizeof(cfl) <= ZC_MAX_OUT));
    char *o_out = zc_out ? sl->hd_out : sl->d_out;
    q->cur_in_fmt = in_fmt;
    rc = process_block_device_impl(q, d_iq, n_in, (want & PMR_WANT_PCM) ? o_out + sl->off_pcm : NULL,
                                   (want & PMR_WANT_AUDIO) ? o_out + sl->off_audio : NULL, stride, &ns,
                                   (want & PMR_WANT_CHAN) ? (zc_out ? sl->hd_chan : sl->d_chan) : NULL,
                                   (want & PMR_WANT_RSSI) ? o_out : NULL, single, phase);
    q->cur_in_fmt = 0;
    if (rc) return rc;
    sl->ns = ns; sl->stride = stride; sl->want = want;
    q->in_block = 1;                              /* the block's state has advanced: losing its outputs now poisons the handle (slot_submit_end) */
    if ((rc = slot_snapshot_mask(q, sl))) return rc;
    if (ns && !zc_out) {
        const size_t lo = (want & PMR_WANT_RSSI) ? 0 : (want & PMR_WANT_PCM) ? sl->off_pcm : sl->off_audio;
        const size_t hi = (want & PMR_WANT_AUDIO) ? sl->off_audio + n * sizeof(float)
                        : (want & PMR_WANT_PCM) ? sl->off_pcm + n * sizeof(int16_t) : (size_t)q->M * sizeof(float);
        if (hi > lo && (want & (PMR_WANT_RSSI | PMR_WANT_PCM | PMR_WANT_AUDIO)))
            HIPCHK(hipMemcpyAsync(sl->h_out + lo, sl->d_out + lo, hi - lo, hipMemcpyDeviceToHost, q->stream), "D2H");
        if (want & PMR_WANT_CHAN) HIPCHK(hipMemcpyAsync(sl->h_chan, sl->d_chan, n * sizeof(cfl), hipMemcpyDeviceToHost, q->stream), "D2H chan");
    }
    HIPCHK(hipEventRecord(sl->done, q->stream), "record");
    q->in_block = 0;
    return PMR_OK;
}

/* wait for the slot's block and spread its compact rows into the caller's [M][pcm_stride] arrays */
static int slot_collect(pmr_chain q, unsigned i, int16_t *pcm, float *audio, unsigned pcm_stride, unsigned *n_frames,
                        pmr_cf32 *chan_out, float *rssi_db)
{
    pmr_slot *sl = &q->slot[i];
    HIPCHK(hipEventSynchronize(sl->done), "wait block");
    const unsigned ns = sl->ns, M = q->M;
    if (n_frames) *n_frames = ns;
    if (ns > pcm_stride && (pcm || audio || chan_out)) return fail(q, PMR_ERANGE, "stride < frames", hipSuccess);
    if (ns) {
        const int16_t *hp = (const int16_t *)(sl->h_out + sl->off_pcm);
        const float *ha = (const float *)(sl->h_out + sl->off_audio);
        for (unsigned k = 0; k < M; k++) {
            const int open = !sl->masked || sl->open_rows[k];   /* closed channel: its pcm / audio rows stay as the caller left them */
            if (open && pcm && (sl->want & PMR_WANT_PCM)) memcpy(pcm + (size_t)k * pcm_stride, hp + (size_t)k * sl->stride, (size_t)ns * sizeof(int16_t));
            if (open && audio && (sl->want & PMR_WANT_AUDIO)) memcpy(audio + (size_t)k * pcm_stride, ha + (size_t)k * sl->stride, (size_t)ns * sizeof(float));
            if (chan_out && (sl->want & PMR_WANT_CHAN)) memcpy((cfl *)chan_out + (size_t)k * pcm_stride, sl->h_chan + (size_t)k * sl->stride, (size_t)ns * sizeof(cfl));
        }
        if (rssi_db && (sl->want & PMR_WANT_RSSI)) memcpy(rssi_db, sl->h_out, (size_t)M * sizeof(float));
    }
    return PMR_OK;
}

int pmr_chain_process_block_f32(pmr_chain q, const pmr_cf32 *iq, unsigned n_in, int16_t *pcm, float *audio,
                                unsigned pcm_stride, unsigned *n_frames, pmr_cf32 *chan_out, float *rssi_db)
{
    return pmr_chain_process_block_fmt(q, iq, 0, n_in, pcm, audio, pcm_stride, n_frames, chan_out, rssi_db);
}

/* the synchronous call on the receiver's own sample format (include/pmr_io.h: 0 cf32, 1 int16, 2 uint8 -- the reference's radio
 * is an RTL-SDR, README.md:12, whose native samples are uint8 pairs that SoapySDR widens to the cf32 of readStream, src/shared.c:62) */
int pmr_chain_process_block_fmt(pmr_chain q, const void *iq, int iq_format, unsigned n_in, int16_t *pcm, float *audio,
                                unsigned pcm_stride, unsigned *n_frames, pmr_cf32 *chan_out, float *rssi_db)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (q->n_inflight) return fail(q, PMR_EINVAL, "collect the submitted blocks first", hipSuccess);
    /* capacity is checked against the closed-form plan BEFORE any state is advanced */
    unsigned ny_plan = 0, ns_plan = 0;
    plan_counts(q, n_in, &ny_plan, &ns_plan);
    if (n_frames) *n_frames = ns_plan;
    if (ns_plan > pcm_stride && (pcm || audio || chan_out)) return fail(q, PMR_ERANGE, "stride < frames", hipSuccess);
    const unsigned want = ((pcm || audio) ? PMR_WANT_PCM : 0) | (audio ? PMR_WANT_AUDIO : 0) | (chan_out ? PMR_WANT_CHAN : 0) |
                          (rssi_db ? PMR_WANT_RSSI : 0);
    int rc = slot_submit(q, 0, iq, iq_format, n_in, want, 1, 0);
    q->in_block = 0;
    if (rc) return rc;
    rc = slot_collect(q, 0, pcm, audio, pcm_stride, n_frames, chan_out, rssi_db);      /* waits for the block's last copy */
    if (rc) return rc;
    if (q->prof_on) prof_resolve(q);
    return PMR_OK;
}

/* Two-step synchronous form: the reference decides the squelch on THIS block's channelizer output (:828-874) before it
 * demodulates the block (:876-906).  pmr_chain_channelize_block runs the block up to the channelizer / discriminator / RSSI and
 * returns; the caller updates the channel mask; pmr_chain_demodulate_block runs the audio part of that block for the channels
 * open NOW.  Together they produce what pmr_chain_process_block_f32 produces with the same mask. */
int pmr_chain_channelize_block(pmr_chain q, const pmr_cf32 *iq, unsigned n_in, unsigned *n_frames, pmr_cf32 *chan_out,
                               unsigned chan_stride, float *rssi_db)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (q->n_inflight) return fail(q, PMR_EINVAL, "collect the submitted blocks first", hipSuccess);
    unsigned ny_plan = 0, ns_plan = 0;
    plan_counts(q, n_in, &ny_plan, &ns_plan);
    if (n_frames) *n_frames = ns_plan;
    if (ns_plan > chan_stride && chan_out) return fail(q, PMR_ERANGE, "stride < frames", hipSuccess);
    const unsigned want = (chan_out ? PMR_WANT_CHAN : 0) | (rssi_db ? PMR_WANT_RSSI : 0);
    int rc = slot_submit(q, 0, iq, 0, n_in, want, 1, 1);
    q->in_block = 0;
    if (rc) return rc;
    return slot_collect(q, 0, NULL, NULL, chan_stride, n_frames, chan_out, rssi_db);
}

int pmr_chain_demodulate_block(pmr_chain q, int16_t *pcm, float *audio, unsigned pcm_stride, unsigned *n_frames)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (q->faulted) return refuse_faulted(q);
    if (!q->pend_audio) return fail(q, PMR_EINVAL, "no channelized block is waiting for its audio part", hipSuccess);
    const unsigned ns = q->pend_audio_ns;
    if (n_frames) *n_frames = ns;
    if (ns > pcm_stride && (pcm || audio)) return fail(q, PMR_ERANGE, "stride < frames", hipSuccess);
    pmr_slot *sl = &q->slot[0];
    int rc = slot_prepare(q, 0, 0);
    if (rc) return rc;
    const unsigned want = ((pcm || audio) ? PMR_WANT_PCM : 0) | (audio ? PMR_WANT_AUDIO : 0);
    const unsigned stride = ns ? (ns + 7u) & ~7u : 8u;
    const size_t n = (size_t)q->M * stride;
    sl->off_pcm = SLOT_ALIGN((size_t)q->M * sizeof(float));
    sl->off_audio = sl->off_pcm + SLOT_ALIGN(n * sizeof(int16_t));
    const size_t out_hi = (want & PMR_WANT_AUDIO) ? sl->off_audio + n * sizeof(float) : sl->off_pcm + n * sizeof(int16_t);
    const int zc_out = !q->sw.no_zerocopy && sl->hd_out && out_hi <= ZC_MAX_OUT;
    char *o_out = zc_out ? sl->hd_out : sl->d_out;
    q->pend_audio = 0;
    q->in_block = 1;                              /* the audio part advances the detector / follow-on filters: an error in it poisons the handle */
    rc = ns ? audio_part(q, q->pend_audio_frame0, ns, (want & PMR_WANT_PCM) ? o_out + sl->off_pcm : NULL,
                         (want & PMR_WANT_AUDIO) ? o_out + sl->off_audio : NULL, stride) : PMR_OK;
    sl->ns = ns; sl->stride = stride; sl->want = want;
    if (!rc) rc = slot_snapshot_mask(q, sl);
    if (!rc && ns && !zc_out && want) {
        hipError_t e_ = hipMemcpyAsync(sl->h_out + sl->off_pcm, sl->d_out + sl->off_pcm, out_hi - sl->off_pcm, hipMemcpyDeviceToHost, q->stream);
        if (e_ != hipSuccess) rc = fail(q, PMR_EHIP, "D2H", e_);
    }
    if (!rc) { hipError_t e_ = hipEventRecord(sl->done, q->stream); if (e_ != hipSuccess) rc = fail(q, PMR_EHIP, "record", e_); }
    q->in_block = 0;
    if (rc) return rc;
    rc = slot_collect(q, 0, pcm, audio, pcm_stride, n_frames, NULL, NULL);
    if (rc) return rc;
    if (q->prof_on) prof_resolve(q);
    return PMR_OK;
}

/* asynchronous pair: up to PIPE_DEPTH blocks between submit and collect */
int pmr_chain_submit_block_fmt(pmr_chain q, const void *iq, int iq_format, unsigned n_in, unsigned want)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (q->n_inflight >= PIPE_DEPTH) return fail(q, PMR_ERANGE, "PIPE_DEPTH blocks already in flight: collect one first", hipSuccess);
    const unsigned i = (q->slot_head + q->n_inflight) % PIPE_DEPTH;
    int rc = slot_submit(q, i, iq, iq_format, n_in, want ? want : PMR_WANT_PCM, !q->overlap, 0);
    q->in_block = 0;
    if (rc) return rc;
    q->n_inflight++;
    return PMR_OK;
}

int pmr_chain_submit_block(pmr_chain q, const pmr_cf32 *iq, unsigned n_in, unsigned want)
{
    return pmr_chain_submit_block_fmt(q, iq, 0, n_in, want);
}

int pmr_chain_collect_block(pmr_chain q, int16_t *pcm, float *audio, unsigned pcm_stride, unsigned *n_frames,
                            pmr_cf32 *chan_out, float *rssi_db)
{
    if (!q) return PMR_EINVAL;
    HIPCHK(hipSetDevice(q->device), "hipSetDevice");
    if (!q->n_inflight) return fail(q, PMR_EINVAL, "no block in flight", hipSuccess);
    int rc = slot_collect(q, q->slot_head, pcm, audio, pcm_stride, n_frames, chan_out, rssi_db);
    if (rc == PMR_ERANGE) return rc;                       /* caller may retry with a larger stride: the block stays queued */
    q->slot_head = (q->slot_head + 1) % PIPE_DEPTH;
    q->n_inflight--;
    return rc;
}

unsigned pmr_chain_blocks_in_flight(pmr_chain q) { return q ? q->n_inflight : 0; }
unsigned pmr_chain_max_in_flight(pmr_chain q) { (void)q; return PIPE_DEPTH; }

/* pinned host memory from THIS library's HIP runtime: what the asynchronous copies of submit / collect need.  The
 * allocations are remembered (host range -> address the device sees), so a synchronous call on a SMALL block can let the front
 * end read the caller's buffer in place over the host link instead of waiting for a copy engine first (host_zero_copy). */
#define HOST_REG_MAX 256
static struct { char *h, *d; size_t n; } g_host_reg[HOST_REG_MAX];
static pthread_mutex_t g_host_reg_lock = PTHREAD_MUTEX_INITIALIZER;
static void host_reg_acquire(void) { pthread_mutex_lock(&g_host_reg_lock); }
static void host_reg_release(void) { pthread_mutex_unlock(&g_host_reg_lock); }

void *pmr_host_alloc(size_t bytes)
{
    void *p = NULL, *d = NULL;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return NULL;
    if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess && d) {
        host_reg_acquire();
        for (int i = 0; i < HOST_REG_MAX; i++)
            if (!g_host_reg[i].h) { g_host_reg[i].h = (char *)p; g_host_reg[i].d = (char *)d; g_host_reg[i].n = bytes ? bytes : 16; break; }
        host_reg_release();
    } else {
        (void)hipGetLastError();
    }
    return p;
}

void pmr_host_free(void *p)
{
    if (!p) return;
    host_reg_acquire();
    for (int i = 0; i < HOST_REG_MAX; i++)
        if (g_host_reg[i].h == (char *)p) { g_host_reg[i].h = NULL; g_host_reg[i].d = NULL; g_host_reg[i].n = 0; }
    host_reg_release();
    (void)hipHostFree(p);
}

/* device-visible address of [p, p + bytes) if it lies inside a pmr_host_alloc allocation, else NULL */
static const void *host_zero_copy(const void *p, size_t bytes)
{
    const char *c = (const char *)p, *r = NULL;
    host_reg_acquire();
    for (int i = 0; i < HOST_REG_MAX && !r; i++)
        if (g_host_reg[i].h && c >= g_host_reg[i].h && c + bytes <= g_host_reg[i].h + g_host_reg[i].n) r = g_host_reg[i].d + (c - g_host_reg[i].h);
    host_reg_release();
    return r;
}

int pmr_chain_process_block(pmr_chain q, const pmr_cf32 *iq, unsigned n_in, int16_t *pcm, unsigned pcm_stride,
                            unsigned *n_frames, pmr_cf32 *chan_out, float *rssi_db)
{
    return pmr_chain_process_block_f32(q, iq, n_in, pcm, NULL, pcm_stride, n_frames, chan_out, rssi_db);
}
