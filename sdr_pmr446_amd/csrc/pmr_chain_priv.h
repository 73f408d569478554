/* pmr_chain_priv.h -- what the translation units of the host side share (pmr_chain.c, pmr_chain_plan.c, pmr_chain_frontend.c,
 * pmr_chain_host.c, pmr_chain_aux.c): the handle, the launch / error macros and the few helpers that cross a file boundary.
 * Not installed, not part of the C-ABI (include/pmr_chain.h is).  Round 6 split the 2 450-line pmr_chain.c along its seams; no
 * behaviour changed. */
#ifndef PMR_CHAIN_PRIV_H
#define PMR_CHAIN_PRIV_H

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

#define HIPCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(q, PMR_EHIP, what, e_); } while (0)

#define LAUNCH_ON(st, slot, expr) do { prof_pending pp_; prof_begin(q, (slot), &pp_, (st)); int rc_ = (expr); \
        prof_end(q, &pp_, (st)); if (rc_) return fail(q, PMR_EHIP, pmr_k_names[slot], (hipError_t)rc_); } while (0)
#define LAUNCH(slot, expr) LAUNCH_ON(q->stream, slot, expr)
#define LAUNCH_FE(slot, expr) LAUNCH_ON(q->sfe, slot, expr)

/* helpers shared between the translation units (hidden: not part of the library's exported surface) */
#define PMR_PRIV __attribute__((visibility("hidden")))
PMR_PRIV extern const char *const pmr_k_names[K_COUNT];          /* profile-slot names (pmr_chain_aux.c) */
PMR_PRIV int fail(pmr_chain q, int code, const char *what, hipError_t e);
PMR_PRIV int refuse_faulted(pmr_chain q);
PMR_PRIV int dev_alloc(pmr_chain q, void **p, size_t bytes);
PMR_PRIV int dev_alloc_state(pmr_chain q, void **p, size_t bytes);
PMR_PRIV int dev_upload(pmr_chain q, float **p, const float *src, size_t n);
PMR_PRIV int upload_padded_taps(pmr_chain q, float **p, const float *h, unsigned n);
PMR_PRIV int fir_fft_upload_spectrum(pmr_chain q, float **dst, unsigned N, const float *h, unsigned n);
PMR_PRIV int fir_fft_pick(const struct pmr_chain_s *q, unsigned ns, unsigned nchan, int dual);
PMR_PRIV void prof_begin(pmr_chain q, int slot, prof_pending *pp, hipStream_t st);
PMR_PRIV void prof_push(pmr_chain q, const prof_pending *pp);
PMR_PRIV void prof_end(pmr_chain q, prof_pending *pp, hipStream_t st);
PMR_PRIV void fe_launch_events(pmr_chain q, int slot, int last_on_stream, pmr_launch_events *ev, prof_pending *pp);
PMR_PRIV void prof_resolve(pmr_chain q);
PMR_PRIV int chain_init(pmr_chain q);
PMR_PRIV void plan_core(unsigned num_stages, uint32_t arb_step, unsigned M, uint64_t n_raw, uint32_t arb_phase, unsigned leftover, unsigned n_in, unsigned *ny_out, unsigned *ns_out, uint32_t *phase_out);
PMR_PRIV void plan_counts(const struct pmr_chain_s *q, unsigned n_in, unsigned *ny_out, unsigned *ns_out);
PMR_PRIV int frontend_staged(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out);
PMR_PRIV int frontend_fused(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out);
PMR_PRIV int frontend_two_level(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny_out);
PMR_PRIV int ctcss_run(pmr_chain q, int64_t frame0, unsigned ns, int fir_done );
PMR_PRIV int ring_to_linear(pmr_chain q, void *dst, const void *ring, uint64_t mask, uint64_t pos, size_t n, size_t elem);
PMR_PRIV int audio_part(pmr_chain q, int64_t frame0, unsigned ns, void *d_pcm, void *d_audio, unsigned pcm_stride);
PMR_PRIV int process_block_device_impl(pmr_chain q, const void *d_iq, unsigned n_in, void *d_pcm, void *d_audio, unsigned pcm_stride, unsigned *n_frames, void *d_chan_out, void *d_rssi_db, int single, int phase );

#endif
