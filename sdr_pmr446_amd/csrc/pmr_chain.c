/* pmr_chain.c -- host side of libpmr446_hip.so: the process-one-block entry points of include/pmr_chain.h.
 *
 * Plain C (the reference's host language).  Owns the per-stream device state that the liquid objects of
 * reference include/sdr_pmr446.h:54-82 (struct _proc_chain_t) own on the CPU, keeps the closed-form sample
 * counters on the host (so no device->host sync is ever needed to size a launch), and enqueues the gfx950
 * kernels of pmr_kernels.hip on one HIP stream.  There is NO CPU fallback: without a HIP device
 * pmr_chain_create() fails.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pmr_chain.h"
#include "../data/pmr446_taps.h"
#include "pmr_design.h"
#include "pmr_kernels.h"
#include "pmr_internal.h"

#define FM_HIST_FRAMES 512u     /* >= 376 (HP) + IIR warm-up; also covers 102/100-tap follow-on FIRs */
#define AUX_HIST_FRAMES 128u    /* history of the time-major intermediates behind the HP stage        */
#define ARB_KEEP 16             /* decimated-sample history kept for the 14-tap arbitrary resampler   */
#define PROF_SLOTS 24
#define PIPE_DEPTH 3u             /* blocks in flight: rings hold history + PIPE_DEPTH blocks; block b's front end waits for
                                   the back end of block b - PIPE_DEPTH.  3 lets the front end run back to back: the back end of
                                   block b (channelizer, audio FIR) then always has a front end to run under */

typedef struct { float re, im; } cfl;

enum { K_DC_AGG, K_DC_SCAN, K_DC_APPLY, K_HALFBAND, K_ARB, K_CHANNELIZE, K_RSSI, K_FIR_HP, K_FIR_DE, K_FIR_LP,
       K_FE, K_FE_TILES, K_CHANNELIZE_SMALL, K_FE_L2, K_CT_FIR, K_CT_DC, K_CT_GOERTZEL, K_FE_TILEFIX, K_SPGRAM, K_COUNT };
static const char *k_names[K_COUNT] = { "k_dcblock<agg>", "k_dc_scan", "k_dcblock<apply>", "k_halfband", "k_arb",
                                        "k_channelize (fused256 / pfb_wide + fft_disc / generic)", "k_rssi_finish",
                                        "audio FIR <hp> (k_fir_fft / k_fir_mfma4 / k_fir_pair)", "audio FIR <deemph>",
                                        "audio FIR <lp>", "k_fe_fast (k_frontend)", "k_fe_carry",
                                        "k_channelize_win", "k_fe_level2", "audio FIR <ctcss_lp>",
                                        "(unused)", "k_ct_seg_agg + scan + goertzel + final", "k_fe_carry_tail / k_fe_tilefix", "k_spgram + finish" };

#define ZC_MAX_IN  (1u << 18)            /* zero-copy synchronous calls: samples (above this a copy engine + HBM-speed kernels win) */
#define ZC_MAX_OUT (1u << 20)            /* ... and bytes of [rssi | pcm | audio] */

typedef struct { hipEvent_t a, b; int slot; } prof_pending;

/* one block in flight between host buffers (pmr_chain_submit_block / _collect_block; the synchronous entry points use slot 0) */
typedef struct {
    cfl *d_in; char *d_out; cfl *d_chan;         /* device: input staging; [rssi | pcm | audio], compact [M][stride]; tap-off */
    void *d_raw;                                 /* device: int16 / uint8 input before conversion (submit_block_fmt)           */
    hipEvent_t in_ready; int used; unsigned par; /* input copy finished; pipeline parity of the block that last used the slot  */
    char *h_out; cfl *h_chan;                    /* pinned host copies of the outputs                                         */
    char *hd_out; cfl *hd_chan;                  /* the same pinned buffers as the DEVICE sees them (zero-copy outputs of small blocks) */
    size_t out_bytes, off_pcm, off_audio;
    hipEvent_t done; unsigned ns, stride, want;
    uint8_t *open_rows; int masked;              /* the channel mask the block's audio part ran under (rows of closed channels are
                                                    never handed to the caller: include/pmr_chain.h, pmr_chain_set_channel_mask) */
} pmr_slot;

struct pmr_chain_s {
    pmr_chain_cfg cfg;
    pmr_design d;
    int device;
    hipStream_t stream;              /* back-end stream (channelizer, audio, outputs): what callers synchronise on */
    hipStream_t stream_fe;           /* front-end stream: block b+1's front end overlaps block b's back end        */
    hipStream_t sfe;                 /* stream the CURRENT call's front end is queued on: stream_fe (pipelined) or stream (single-stream
                                        calls: the synchronous host entry point and set_overlap(0) -- no cross-stream events at all)   */
    int last_single;                 /* the previous call was a single-stream one                                   */
    hipEvent_t ev_switch;            /* orders stream_fe behind stream when a pipelined call follows a single-stream one */
    hipEvent_t input_ready; int has_input_ready;   /* caller's "d_iq is complete" event for the NEXT device-entry call */
    hipStream_t stream_h2d;          /* input copies of the asynchronous host-buffer pair: H2D of block b+1 under the kernels of block b */
    hipEvent_t ev_fe[PIPE_DEPTH], ev_be[PIPE_DEPTH];   /* front end / back end of block (n mod PIPE_DEPTH) finished     */
    /* CTCSS detector of pipelined calls on a stream of its own: four launch-latency-bound kernels that only the NEXT block's
     * detector waits for -- behind them on the back-end stream, the next block's carry / channelizer / FIR waited too */
    hipStream_t stream_ct; hipEvent_t ev_ct[PIPE_DEPTH], ev_ctlp; int ct_ev_used[PIPE_DEPTH], ct_async_last; unsigned ct_last_par, cur_par; int cur_single;
    int overlap;                     /* two-stream pipelining enabled (PMR_OVERLAP=0 disables)                     */
    int fe_prio_high;                /* the front-end stream was created at high priority (two-level plan + FFT form of the audio FIR) */
    uint64_t n_calls;
    unsigned M, res_size, chan_size;
    char err[256];

    /* constant tables on the device */
    float *d_hb_h1[PMR_MAX_STAGES];
    float *d_arb_bank, *d_pfb_taps_t, *d_fft_tw, *d_nco_cs, *d_lam_thread_pow, *d_lam_tile_idx_pow;
    float *d_hp_pad, *d_lp_pad, *d_de_pad;
    unsigned hp_len, lp_len, de_len;
    /* overlap-save FFT form of the audio FIR (pmr_fir_fft.hip): device tables per transform size (0: 1024, 1: 4096 points) */
    int fft_ok; pmr_fir_fft_tab fft_tab[3]; float *d_fft_H[3], *d_fft_H2[3], *d_fft_TA[3], *d_fft_TB[3];   /* per transform size (0: 1024, 1: 4096, 2: 2048 points) */
    pmr_dc_consts dcc;

    /* carried state / work buffers on the device */
    cfl *d_in;                       /* staging for host blocks [max_block]                  */
    cfl *d_dc_state, *d_dc_agg, *d_dc_W;
    cfl *d_z[PMR_MAX_STAGES + 1];    /* z_0 .. z_h, each [keep | new]                         */
    unsigned keep[PMR_MAX_STAGES + 1];
    cfl *d_xr; uint64_t xr_mask;      /* resampled ring, sample a at d_xr[a & xr_mask]                 */
    float *d_fm, *d_aux1, *d_aux2;   /* row rings, frame t at ring[(t & fm_mask) * M + k]             */
    uint64_t fm_mask;
    void *d_scratch; size_t scratch_bytes;
    float *d_rssi_part;
    int faulted, in_block;                       /* PMR_EHIP inside a block: no further blocks until pmr_chain_reset */
    pmr_rssi_job rssi_job; int rssi_job_pending; /* RSSI finish of the block in hand, waiting to ride in the audio FIR's launch */
    size_t rssi_part_cap;
    pmr_slot slot[PIPE_DEPTH]; unsigned slot_head, n_inflight;

    /* open-channel mask (reference :876-877) and per-channel discriminator reset (:866) */
    unsigned *d_chan_list; unsigned n_enabled; int mask_on; uint8_t *h_open;   /* h_open[k]: channel k enabled (host copy of the mask) */
    uint8_t *d_reset_flags, *h_reset_flags; int reset_pending;

    /* CTCSS branch (pmr_ctcss.hip), allocated by pmr_chain_ctcss_enable */
    int ct_on; unsigned ct_max_ev, ct_nev_last; int ct_sel;
    float *d_ct_taps_ext;            /* the low-pass-branch taps zero-extended to the folded audio filter's length (dual pass) */
    float *d_ctlp, *d_ct_taps, *d_ct_lampow, *d_ct_agg, *d_ct_W, *d_ct_dcstate, *d_ct_U, *d_ct_coef, *d_ct_part, *d_ct_carry[2];
    pmr_ctcss_event *d_ct_events;
    uint8_t *d_ct_restart;           /* [M] 1: the channel's Goertzel sums were restarted inside the block in progress (reset / opened):
                                        that block's event is reported as "no decision" (k_ct_final clears the flag)              */
    uint8_t *ct_open_last; int ct_masked_last;   /* the mask the LAST block's detector ran under (pmr_chain_ctcss_read)           */
    unsigned hp_len_raw;             /* length of the un-folded high-pass table (377)                 */

    /* fused front end (pmr_frontend.hip): geometry, gain tables, raw history, dc probes */
    pmr_switches sw;                 /* A/B switches, read once from the environment at create */
    int chan_small;                  /* small-M channelizer (pmr_channelize_small.hip) selected       */
    int chan_wide;                   /* wide-bank channelizer (pmr_channelize_wide.hip: filter bank + radix-4 FFT kernels) */
    cfl *d_chan_x;                   /* its scratch: polyphase bank outputs [chan_size + 1][M]         */
    int fe_on, fe_nt, fe_spt;        /* fused path selected; threads per tile workgroup, samples per thread */
    unsigned fe_lds_pad;             /* pmr_fe_params.lds_pad of this plan (chain_init)                */
    int fe_fast_fmt;                 /* the plan's front-end kernel converts int16 / uint8 input as it loads (k_fe_fast, 256 x 16 tiles) */
    int cur_in_fmt;                  /* sample format of THIS call's d_iq (0 cf32): set by the synchronous zero-copy path of slot_submit */
    int fe_T_own, fe_Hh, fe_HhQ, fe_TQ, fe_hcap;
    int fe_m[PMR_FE_MAX_STAGES], fe_tap_off[PMR_FE_MAX_STAGES];
    float fe_Kgain, fe_lam_wave, fe_lam_pow16[6];
    float fe_taps_host[PMR_FE_MAX_STAGES * 64];
    float *d_fe_taps, *d_fe_GA, *d_fe_T1, *d_fe_T2, *d_fe_lam_lane;
    cfl *d_fe_hist[2], *d_fe_vstate[2], *d_fe_probeA, *d_fe_probeB, *d_fe_probeL, *d_fe_probeE, *d_fe_V[PIPE_DEPTH];
    /* two-level front end for deep cascades: level 1 = dc-block + first fe_s1 stages -> decimated ring, level 2 = rest */
    int fe_two;                      /* 1: two launches of k_frontend (modes 1 and 2)                 */
    int fe_s1;                       /* stages in level 1 (all 6-tap)                                  */
    int fe2_T_own, fe2_Hh, fe2_HhQ, fe2_TQ;   /* level-2 tile geometry, in level-1 output samples      */
    int fe2_N0, fe2_fast;            /* level-2 tile size; specialised k_fe_level2<MA, MB> selected       */
    float fe1_K;                     /* alpha * prod G_e (e < s1): dc-carry gain at the level-1 output */
    float *d_fe_G1;                  /* [..] fe1_K * mu^e: level 1's carry gain per tile-local index (level 2's load-time fix) */
    cfl *d_fe_ring1; uint64_t ring1_mask;
    uint64_t *d_fe_tile_j; float *d_fe_rho_pow; unsigned fe_K;   /* k_fe_tilefix inputs */
    /* dc carry applied where the channelizer loads the resampled stream (pmr_carry_fix): supported by this plan's kernels;
     * used by THIS call; table mu^q' as one float product; LDS table length; decimated samples per frame; index bias (tiles) */
    int cal_ok, cal_now; float *d_fe_G12, *d_fe_GAK; unsigned cal_nv, cal_adv_q, cal_nbias;
    unsigned cal_fix_limit, cal_ntiles, cal_slot; uint32_t cal_phi0;   /* ... of this call's block (frontend_fused) */
    int tf_on_backend, pend_tf; unsigned pend_tf_Q;    /* one-level form: k_fe_tilefix deferred to the back-end stream (uses pend_t2 / pend_f2) */
    int tf_last_be;                  /* the previous pipelined call's carry pass ran on the back-end stream */
    /* two-step synchronous form (pmr_chain_channelize_block / _demodulate_block): the audio part of the block channelized last */
    int pend_audio; int64_t pend_audio_frame0; unsigned pend_audio_ns;
    int l2_on_backend, pend_l2; pmr_fe_params pend_p2; pmr_fe_tiles_params pend_t2; pmr_fe_fix_params pend_f2; unsigned pend_ntiles2;
    int fe_sel;                      /* which of the ping-pong history / state buffers is current     */
    /* waterfall periodogram (pmr_spectrum.hip): display width (0 = off), window / twiddle tables, per-workgroup partial rows, PSD */
    unsigned spec_nfft, spec_ntr_last; float *d_spec_win, *d_spec_tw, *d_spec_part, *d_spec_psd;
    unsigned fe_max_tiles;

    /* host-side counters (all closed form in the number of samples consumed) */
    uint64_t n_raw;                  /* raw samples consumed since reset                      */
    uint32_t arb_phase;              /* resamp_crcf phase, 2^24 per decimated sample          */
    uint64_t xr_abs;                 /* resampled samples produced since reset                */
    uint64_t frames_done;            /* frames channelized since reset                        */
    unsigned last_ny, last_ns;
    int dbg_on; cfl *d_dbg_xr; float *d_dbg_fm, *d_dbg_ct;

    /* profiling */
    int prof_on; unsigned prof_tick;
    hipEvent_t fe_done_ev; int fe_done_used;   /* event the front-end stream's LAST launch of this call signals itself (pipelined
                                                  calls: ev_fe[par]) and whether a launch took it */
    double prof_ms[PROF_SLOTS]; unsigned prof_n[PROF_SLOTS];
    prof_pending *pend; unsigned npend, cappend;
    hipEvent_t *pool; unsigned npool, cappool;
};

/* ------------------------------------------------------------------------------------------- */

static int fail(pmr_chain q, int code, const char *what, hipError_t e)
{
    if (q) snprintf(q->err, sizeof(q->err), "%s%s%s", what, e != hipSuccess ? ": " : "",
                    e != hipSuccess ? hipGetErrorString(e) : "");
    /* ANY failure in the middle of a block (q->in_block: between the first launch / counter update of a block and its last) leaves
     * the stream position undefined -- the host counters may be ahead of what the device did, whether the cause was the HIP runtime,
     * an allocation or a capacity check that could only be made mid-way: the handle refuses further blocks until pmr_chain_reset */
    if (q && q->in_block) q->faulted = 1;
    return code;
}

static int refuse_faulted(pmr_chain q)
{
    return fail(q, PMR_EHIP, "an earlier block failed mid-way: the stream position is undefined, call pmr_chain_reset", hipSuccess);
}

#define HIPCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(q, PMR_EHIP, what, e_); } while (0)

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
static int dev_alloc(pmr_chain q, void **p, size_t bytes) { return dev_alloc_fill(q, p, bytes, pmr_debug_poison_enabled() ? 0xFF : 0); }
static int dev_alloc_state(pmr_chain q, void **p, size_t bytes) { return dev_alloc_fill(q, p, bytes, 0); }

static int dev_upload(pmr_chain q, float **p, const float *src, size_t n)
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
static int upload_padded_taps(pmr_chain q, float **p, const float *h, unsigned n)
{
    size_t len = n + 2 * PMR_TAP_PAD;
    float *tmp = (float *)calloc(len, sizeof(float));
    if (!tmp) return fail(q, PMR_ENOMEM, "calloc", hipSuccess);
    for (unsigned j = 0; j < n; j++) tmp[j + PMR_TAP_PAD] = h[j];
    int rc = dev_upload(q, p, tmp, len);
    free(tmp);
    return rc;
}

/* Tables of the FFT form of the audio FIR for the folded tap set g[n] (pmr_fir_fft.hip): spectra in the kernel's position order and
 * exact twiddles, both transform sizes.  h2 != NULL: the second tap set of the DUAL pass (CTCSS low-pass branch), zero-extended to n. */
static int fir_fft_upload_spectrum(pmr_chain q, float **dst, unsigned N, const float *h, unsigned n)
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
static int fir_fft_pick(const struct pmr_chain_s *q, unsigned ns, unsigned nchan, int dual)
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

/* ---- profiling helpers: HIP events on the chain's stream around every launch ---- */
static void prof_begin(pmr_chain q, int slot, prof_pending *pp, hipStream_t st)
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

static void prof_push(pmr_chain q, const prof_pending *pp)
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

static void prof_end(pmr_chain q, prof_pending *pp, hipStream_t st)
{
    if (pp->slot < 0) return;
    hipEventRecord(pp->b, st);
    prof_push(q, pp);
}

/* Events a front-end launch carries itself (pmr_launch_events: no packets of their own on the stream).
 *  - profile mode m >= 2: every (m-1)-th launch of the front-end kernel takes a start/stop pair (kernel begin..end);
 *  - otherwise the launch that is the front-end stream's last of this call signals "front end done" (fe_done_ev). */
static void fe_launch_events(pmr_chain q, int slot, int last_on_stream, pmr_launch_events *ev, prof_pending *pp)
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

static void prof_resolve(pmr_chain q)
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

#define LAUNCH_ON(st, slot, expr) do { prof_pending pp_; prof_begin(q, (slot), &pp_, (st)); int rc_ = (expr); \
        prof_end(q, &pp_, (st)); if (rc_) return fail(q, PMR_EHIP, k_names[slot], (hipError_t)rc_); } while (0)
#define LAUNCH(slot, expr) LAUNCH_ON(q->stream, slot, expr)
#define LAUNCH_FE(slot, expr) LAUNCH_ON(q->sfe, slot, expr)

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

static int chain_init(pmr_chain q)
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

/* Closed-form sample accounting for a block of n_in raw samples (no device round trip):
 *   decimated samples Q = floor((n_raw+n_in)/D) - floor(n_raw/D)      (msresamp buffer_index rule)
 *   resampled outputs ny from the 24-bit phase accumulator            (resamp_crcf)
 *   frames ns = floor((leftover + ny) / M)                            (ring rule, :804)             */
static void plan_core(unsigned num_stages, uint32_t arb_step, unsigned M, uint64_t n_raw, uint32_t arb_phase,
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

static void plan_counts(const struct pmr_chain_s *q, unsigned n_in, unsigned *ny_out, unsigned *ns_out)
{
    plan_core(q->d.num_stages, q->d.arb_step, q->M, q->n_raw, q->arb_phase,
              (unsigned)(q->xr_abs - q->frames_done * q->M), n_in,
              ny_out, ns_out, NULL);
}

/* ------------------------------------------------------------------------------------------- */
/* front end, staged: dc-block (:795) -> half-band cascade -> arbitrary resampler (:796)         */

static int frontend_staged(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out)
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
static int frontend_fused(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out)
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
static int frontend_two_level(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out)
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

/* ------------------------------------------------------------------------------------------- */

static int ring_to_linear(pmr_chain q, void *dst, const void *ring, uint64_t mask, uint64_t pos, size_t n, size_t elem);

/* ---- CTCSS branch: low-pass branch FIR -> dc-block scan -> Goertzel bank, all channels ---- */
static int ctcss_run(pmr_chain q, int64_t frame0, unsigned ns, int fir_done /*the low-pass branch is already in d_ctlp*/)
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

/* copy `n` elements starting at absolute ring index `pos` into a linear device buffer (debug capture) */
static int ring_to_linear(pmr_chain q, void *dst, const void *ring, uint64_t mask, uint64_t pos, size_t n, size_t elem)
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
static int audio_part(pmr_chain q, int64_t frame0, unsigned ns, void *d_pcm, void *d_audio, unsigned pcm_stride)
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
        if (rd > 0) return fail(q, PMR_EHIP, k_names[K_FIR_HP], (hipError_t)rd);
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
static int process_block_device_impl(pmr_chain q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio,
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

int pmr_chain_profile_enable(pmr_chain q, int on) { if (!q) return PMR_EINVAL; q->prof_on = on; return PMR_OK; }

int pmr_chain_set_overlap(pmr_chain q, int on)
{
    if (!q) return PMR_EINVAL;
    int rc = pmr_chain_synchronize(q);
    q->overlap = on ? 1 : 0;
    return rc;
}

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
const char *pmr_chain_profile_name(pmr_chain q, unsigned i) { (void)q; return i < K_COUNT ? k_names[i] : NULL; }

int pmr_chain_profile_get(pmr_chain q, unsigned i, double *total_ms, unsigned *launches)
{
    if (!q || i >= K_COUNT) return PMR_EINVAL;
    hipStreamSynchronize(q->stream);
    prof_resolve(q);
    if (total_ms) *total_ms = q->prof_ms[i];
    if (launches) *launches = q->prof_n[i];
    return PMR_OK;
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
