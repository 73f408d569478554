/* pmr_kernels.h -- C-callable launchers of the gfx950 kernels (internal to libpmr446_hip.so).
 * All pointers are device pointers; every launcher enqueues on `stream` and returns the hipError_t
 * of the launch as int (0 == hipSuccess).  Index conventions: tests/chain_model.py. */
#ifndef PMR_KERNELS_H
#define PMR_KERNELS_H

#include <stdint.h>

#include "pmr_experiment.h"             /* the gate of every compile-time experiment hook: first, before any hook's default */
#include "../../include/pmr_chain.h"     /* pmr_ctcss_event */

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pmr_stream_t;     /* hipStream_t */

/* Environment switches: read ONCE per handle by pmr_chain_create; nothing on a launch path calls getenv.
 * All zero = the product. */
typedef struct {
    int fir_direct;         /* PMR_FIR=direct: the direct (MFMA) form of the audio FIR for every block (default: FFT form for large blocks) */
    int fir_fft1024, fir_fft2048;   /* PMR_FIR=fft1024 / fft2048: force the 1024- / 2048-point kernels (the plan picks between the two by the call's
                               frame count, fir_fft_pick) -- A/B and tests */
    int fir_fft4096;        /* PMR_FIR=fft4096: the FFT form's 4096-point kernels (256 threads, 35 / 71 KB of LDS) wherever the FFT form runs
                               -- the variant that lost its A/B in the chain (r04_ab_log.txt) but is compiled in: tests run it */
    int no_overlap;         /* PMR_OVERLAP=0: single-stream calls (also pmr_chain_set_overlap)                */
    int no_zerocopy;        /* PMR_ZEROCOPY=0: synchronous host calls always go through the copy engines (H2D / D2H) */
    int carry_inplace;      /* PMR_CARRY=inplace: one-level front end's dc carry by the read-modify-write pass over the whole block
                               (k_fe_tilefix) instead of at the channelizer's loads -- the form the at-load one must equal bit for bit */
} pmr_switches;

#define PMR_DC_TILE 4096u       /* raw samples per dc-block tile (256 threads x 16) */
#define PMR_DC_SCAN_THREADS 1024u
#define PMR_TAP_PAD 64u         /* zeros on both sides of every FIR tap table handed to the FIR kernels */
#define PMR_AUDIO_R 32u         /* outputs per thread in the time-major FIR */
#define PMR_AUDIO_J 6u          /* IIR warm-up outputs (de-emphasis pole^J < 1e-10); R+J even (packed FMA) */

/* constants of the dc blocker, filled by the host from pmr_design */
typedef struct {
    float a1;                   /* -1 + alpha */
    float lam_pow16[8];         /* lambda^(16 * 2^j), j = 0..7 : thread-chunk scan multipliers      */
    float lam_tile_pow[10];     /* (lambda^4096)^(2^j), j = 0..9 : tile scan multipliers             */
} pmr_dc_consts;

/* dc blocker, reference src/sdr_pmr446.c:795.
 *  pass 0: tile aggregates   agg[tile] = sum_i lambda^(4095-i) x[tile*4096+i]   (zeros beyond n_in)
 *  scan  : W[tile] = v just before the tile; *state <- v after the last valid sample
 *  pass 1: yb[n] written to out[n]                                                                  */
int pmr_launch_dc_agg(pmr_stream_t s, const void *x, unsigned n_in, void *agg, const pmr_dc_consts *c,
                      const float *lam_thread_pow /*[256] lambda^(16 t)*/);
int pmr_launch_dc_scan(pmr_stream_t s, const void *agg, unsigned ntiles, void *W, void *state,
                       const pmr_dc_consts *c, const float *lam_tile_idx_pow /*[1024] (lambda^4096)^t*/,
                       float lam_last /*lambda^L_last*/, float inv_last /*lambda^-(4096-L_last)*/);
int pmr_launch_dc_apply(pmr_stream_t s, const void *x, unsigned n_in, const void *W, void *out,
                        const pmr_dc_consts *c, const float *lam_thread_pow);

/* one half-band decimation stage (:796).  zin[keep_in + r] is the sample r positions after the
 * first NEW input sample (r < 0: history); par = parity of the absolute index of that first new sample.    */
int pmr_launch_halfband(pmr_stream_t s, const void *zin, void *zout, unsigned n_out, int keep_in, int par,
                        int m, const float *h1, float scale);

/* arbitrary polyphase resampler (:796).  dec[keep + q] = q-th new decimated sample.                  */
int pmr_launch_arb(pmr_stream_t s, const void *dec, void *out_ring, uint64_t out_pos0, uint64_t out_mask,
                   unsigned ny, uint32_t phase0, uint32_t step, const float *bank, int keep);

/* int16 (fmt 1, x / 32768) or uint8 (fmt 2, (x - 127.5) / 127.5) interleaved I/Q -> cf32 on the device (include/pmr_io.h rules) */
int pmr_launch_iq_convert(pmr_stream_t s, const void *raw, void *out_cf32, unsigned n_in, int fmt);

/* Ring buffers.  The resampled stream lives in a power-of-two ring of cf32 addressed by the ABSOLUTE resampled
 * sample index (sample a at xr[a & xr_mask]); the discriminator output in a ring of time-major rows addressed by the
 * absolute frame index (frame t at fm[(t & fm_mask) * M + k]).  Frame f covers samples [f*M, (f+1)*M); the NCO phase
 * of sample a is table entry a mod period.  Negative indices (before the stream start) wrap onto still-zero memory. */
/* DC carry applied WHERE THE CHANNELIZER LOADS the resampled stream (one-level front ends: cfg2, cfg3).  The front end's tiles run
 * their dc blocker from zero state; the missing carry adds V_c * K * GA[branch] * mu^q' to the outputs of tile c (pmr_frontend.hip).
 * Instead of a read-modify-write pass over the whole resampled block (k_fe_tilefix: 2 x 8 x rate bytes per input sample), the
 * carry pass only computes the V_c and corrects IN PLACE the block's last few outputs [fix_limit, ny) -- the part later calls
 * re-read as filter history -- and the channelizer subtracts the same term, in the same arithmetic (bit for bit), from every
 * sample j < fix_limit as it loads it.  V == NULL: nothing to do at load (the ring holds corrected samples). */
typedef struct {
    const void *V;              /* [ntiles] cf32 carries of this block's front-end tiles                          */
    const float *GA, *G12;      /* [256] Kgain * gain per polyphase branch (one float product); [TQ + HhQ + 32] mu^q' (= T1[q' >> 5] * T2[q' & 31]) */
    uint64_t pos0;              /* absolute ring index of the block's first output                                */
    uint32_t phi0, step;        /* resampler phase before the block's first decimated sample, step (2^24 per decimated sample) */
    uint32_t fix_limit;         /* outputs j < fix_limit are corrected at load                                    */
    uint32_t ntiles, TQ, HhQ;   /* tiles of the block; decimated samples a tile owns; its halo in decimated samples */
    uint32_t qbias, nbias;      /* nbias * TQ: added to decimated indices so history before the block stays non-negative */
    uint32_t nv;                /* carries one workgroup can meet (LDS table length)                              */
} pmr_carry_fix;

typedef struct {
    const void *xr; uint64_t xr_mask;   /* resampled ring                                                   */
    int64_t frame0;                     /* absolute index of the first NEW frame of this call               */
    uint64_t xr_end;                    /* absolute index one past the last valid resampled sample          */
    float *fm; uint64_t fm_mask;        /* discriminator ring (rows)                                        */
    unsigned ns, M, p;
    const float *taps_t, *fft_tw, *nco_cs; unsigned nco_period;
    float fm_ref;
    void *chan_out; unsigned chan_stride;   /* nullable, channel-major [M][chan_stride], frame index relative */
    float *rssi_part;                       /* nullable, [ntiles][M] partial sums of |y|                      */
    const uint8_t *reset_flags;             /* nullable, [M]: non-zero = freqdem_reset (:866) before this call's first frame,
                                               i.e. the channel's first discriminator output is arg(0) = 0   */
    pmr_carry_fix fix;                      /* dc carry applied at load (fix.V == NULL: off); honoured by the kernels
                                               pmr_channelize_carry_at_load() names                                       */
} pmr_chan_params;

/* does the channelizer this (M, p, nco_period, switches) selects apply pmr_carry_fix at load?  `adv_q` = decimated samples one
 * frame advances (upper bound), TQ as above */
int pmr_channelize_carry_at_load(unsigned M, unsigned p, unsigned nco_period, int chan_small, int chan_wide,
                                 unsigned adv_q, unsigned TQ);
/* LDS table length (pmr_carry_fix.nv) for that kernel */
unsigned pmr_channelize_carry_nv(unsigned M, unsigned adv_q, unsigned TQ);

/* NCO shift + polyphase analysis bank + M-point FFT + discriminator (:808-821, :881), any power-of-two M */
int pmr_launch_channelize(pmr_stream_t s, const pmr_chan_params *p, unsigned *ntiles_out);
int pmr_launch_rssi_finish(pmr_stream_t s, const float *rssi_part, unsigned ntiles, unsigned M, unsigned ns,
                           float *rssi_db);

/* wide banks (pmr_channelize_wide.hip): M = 256 in one fused kernel; M = 64 / 1024 / 4096: filter-bank kernel parallel over channels,
 * then radix-4 FFT + discriminator kernel parallel over frames; x_scratch holds (ns + 1) * M complex floats.  Same outputs. */
int pmr_channelize_wide_supported(unsigned M, unsigned p, unsigned nco_period);
int pmr_launch_channelize_wide(pmr_stream_t s, const pmr_chan_params *p, void *x_scratch, unsigned *ntiles_out);

/* the 16-channel, 26-tap bank of the PMR446 plan (pmr_channelize_small.hip): staged window, sliding-window bank, FFT in registers.
 * Same outputs as pmr_launch_channelize. */
int pmr_channelize_small_supported(unsigned M, unsigned p, unsigned nco_period);
int pmr_launch_channelize_small(pmr_stream_t s, const pmr_chan_params *p, unsigned *ntiles_out);

/* time-major real FIR with optional epilogue (:882-904).
 *  in        time-major, in[(t)*M + k], t = 0 first new frame (history at negative t)
 *  taps_pad  [ntaps + 2*(R+J-1)]: tap(e, i) = taps_pad[(ntaps + R+J - 2 - e) + i] is the weight of input step e
 *            in accumulator i, i.e. h zero-padded by R+J-1 on both sides, in natural order
 *  gain      multiplies the FIR output (:890)
 *  iir       if non-zero: y = b0*v0 + b1*v1, v0 = u - a1*v1 (:898)
 *  out_tm    nullable time-major output (same indexing as `in`)
 *  pcm/audio nullable channel-major [M][stride] final outputs                                          */
/*  in / out_tm are row rings: frame t (absolute) at ring[(t & row_mask) * M + k]; row0 = absolute index of the
 *  first new frame.  pcm / audio are the caller's channel-major buffers, frame index relative to row0.          */
/*  chan_list / n_chan: open-channel mask (device array of enabled channel indices; NULL = every channel).  Honoured by the
 *  MFMA kernel; the VALU A/B versions compute every channel (rows of disabled channels are then written too).   */
/*  job / job_done: a small reduction that rides in the launch instead of costing a kernel boundary of its own (4.6 us at the
 *  reference's block size): the RSSI finish of the block's channelizer (k_rssi_finish's arithmetic, one extra workgroup).  Taken by
 *  the 16x16x4 MFMA kernel on blocks of a few tiles; *job_done says whether it was (otherwise the caller launches k_rssi_finish). */
typedef struct { const float *rssi_part; unsigned ntiles, M, ns; float *rssi_db; } pmr_rssi_job;
int pmr_launch_fir_tm(pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0, unsigned ns, unsigned M,
                      const float *taps_pad, unsigned ntaps, float gain, int iir, float b0, float b1, float a1,
                      float *out_tm, int16_t *pcm, float *audio, unsigned stride, const unsigned *chan_list, unsigned n_chan,
                      const pmr_rssi_job *job /*nullable*/, int *job_done /*nullable*/);
/* audio FIR (-> pcm / audio) and a second tap set of the same length (-> time-major out2_tm) in ONE pass over the samples;
 * returns -1 when the MFMA kernel cannot take it (caller then runs two passes) */
int pmr_launch_fir_dual(pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0, unsigned ns, unsigned M,
                        const float *taps_pad, const float *taps2_pad, unsigned ntaps, int16_t *pcm, float *audio, unsigned stride,
                        float *out2_tm, const unsigned *chan_list, unsigned n_chan);

/* audio FIR of large blocks by overlap-save FFT convolution (pmr_fir_fft.hip): the same linear filter, ~15x fewer operations.
 * Device tables of one transform size (which: 0 = 1024 points, 1 = 4096): H / H2 = spectra of the tap sets / N in the kernel's
 * position order, TA / TB = twiddles; built on the host in double (pmr_fir_fft_spectrum / _twiddles). */
typedef struct { const float *H, *H2, *TA, *TB; } pmr_fir_fft_tab;
unsigned pmr_fir_fft_size(int which);
void pmr_fir_fft_spectrum(unsigned N, const float *h, unsigned ntaps, float *H_out /*[2 N]*/);
void pmr_fir_fft_twiddles(unsigned N, float *TA /*[15][N/16][2]*/, float *TB /*[16][N/256][2]*/);
int pmr_fir_fft_supported(unsigned M, unsigned ntaps);
int pmr_launch_fir_fft(pmr_stream_t s, int which, const pmr_fir_fft_tab *tab, const float *in, uint64_t row_mask, int64_t row0,
                       unsigned ns, unsigned M, unsigned ntaps, int16_t *pcm, float *audio, unsigned stride,
                       float *out2_tm /*second product, time-major (needs tab->H2); NULL: none*/,
                       const unsigned *chan_list, unsigned n_chan);

/* audio FIR on the matrix pipe, the product (pmr_fir_mfma4.hip): banded-Toeplitz x data with v_mfma_f32_16x16x4_f32, 128-frame
 * tiles; chan_list as below; taps2_pad / out2_tm: optional second tap set -> time-major, same pass */
int pmr_fir_mfma4_supported(unsigned M, unsigned ntaps);
int pmr_launch_fir_mfma4(pmr_stream_t s, const float *in, uint64_t row_mask, int64_t row0, unsigned ns, unsigned M,
                         const float *taps_pad, unsigned ntaps, float *out_tm, int16_t *pcm, float *audio, unsigned stride,
                         const unsigned *chan_list, unsigned n_chan, const float *taps2_pad, float *out2_tm,
                         const pmr_rssi_job *job /*nullable*/, int *job_done /*nullable*/);
/* ---- CTCSS branch (pmr_ctcss.hip) ---- */
#define PMR_CT_TONES 38u
#define PMR_CT_SEG 16u          /* time segments a Goertzel block is split into */
#define PMR_CT_BLOCK 2441u      /* CTCSS_BLOCK_SIZE, src/sdr_pmr446.c:37,:46 */
/* the whole detector behind the low-pass branch `lp` (time-major ring, NOT modified): dc blocker of ctcss_execute (:606) as a
 * scan on the Goertzel segment grid, 38-tone Goertzel bank over N-frame blocks, decision (:366-409).  chan_list / n_chan: the
 * open channels (device array; NULL = all M): the detector runs for those only (reference :893).  lampow[n] = lambda^n, n <= 160;
 * agg / W: [segments][M] work arrays; state: [M] blocker state carried across calls */
int pmr_launch_ct_detector(pmr_stream_t s, const float *lp, uint64_t row_mask, int64_t row0, unsigned ns, unsigned M, unsigned N,
                           float a1, const float *lampow, float *state, float *agg, float *W, const float *U, const float *coef,
                           float *part, const float *carry_in, float *carry_out, pmr_ctcss_event *events,
                           unsigned char *restart /*[M] detector restarted inside the block in progress: its event = no decision*/,
                           unsigned nblk, unsigned ncomplete, const unsigned *chan_list, unsigned n_chan);
unsigned pmr_ct_max_segments(void);      /* segments (16 per Goertzel block) one call of the detector can cover */

/* ---- `dsd_in` back end (pmr_dsd_kernels.hip): discriminator + msresamp_rrrf interpolator on absolute-indexed rings ---- */
int pmr_launch_dsd_fm(pmr_stream_t s, const void *xr, uint64_t xr_mask, uint64_t a0, unsigned ny, float *fm,
                      uint64_t fm_mask, float ref);
int pmr_launch_dsd_arb(pmr_stream_t s, const float *fm, uint64_t fm_mask, uint64_t j0, unsigned nu, uint32_t step,
                       const float *bank, float *u, uint64_t u_mask, int16_t *pcm, float *audio);
int pmr_launch_dsd_hb(pmr_stream_t s, const float *in, uint64_t in_mask, uint64_t i0, unsigned n, int m,
                      const float *h1, float *out, uint64_t out_mask, int16_t *pcm, float *audio);

/* ---- fused front end (pmr_frontend.hip): dc-block + half-band cascade + arbitrary resampler in one pass ---- */
#define PMR_FE_MAX_STAGES 16
typedef struct {
    const void *x;              /* new block [n_in]: cf32, or the raw integer samples when in_fmt != 0 */
    unsigned lds_pad;           /* bytes of unused LDS added to every tile workgroup of the specialised kernels: shapes how front-end tiles
                                   and the back end's workgroups share a CU (pmr_chain.c fe_init: 256-channel plans) */
    int in_fmt;                 /* 0 cf32; 1 interleaved int16 / 32768; 2 interleaved uint8, (x - 127.5) / 127.5 (include/pmr_io.h):
                                   converted as the tile is loaded -- synchronous zero-copy calls on the receiver's own sample format
                                   (k_fe_fast only: pmr_fe_fast_covers; same arithmetic as k_iq_convert)                            */
    const void *hist;           /* raw history: the hcap samples before the block                    */
    void *new_hist;             /* raw history for the NEXT call (other ping-pong buffer), written by tile 0 */
    void *out;                  /* resampled ring; output j of this block goes to out[(out_pos0 + j) & out_mask] */
    uint64_t out_pos0, out_mask;
    void *probeA, *probeB;      /* [ntiles] local dc state at tile offsets Hh-1 and N0-1             */
    void *probeL, *probeE;      /* local dc state at (block start - 1) in tile 0, (block end) in tile c_end */
    const float *hb_taps;       /* branch taps of all stages, execution order, oldest-first          */
    const float *arb_bank;      /* [256][14]                                                         */
    const float *lam_lane_pow;  /* [64] lambda^(spt l)                                               */
    unsigned n_in, ny, Q;       /* raw samples, resampled outputs, decimated samples of this block   */
    uint32_t phi0, step;        /* resamp_crcf phase before the block's first decimated sample, step */
    int h, T_own, Hh, HhQ, TQ;  /* stages; owned raw samples per tile, halo (raw / decimated), owned decimated */
    int pend, hcap, c_end, off_end;
    int mode;                   /* 0 whole front end; 1 level 1 (store decimated stream); 2 level 2 (ring input) */
    /* level 2 input: decimated ring of level 1; in_abs0 = absolute index of this call's first new sample; the dc carry
     * of level-1 tile c1 = j / fix_TQ is subtracted at load: V[c1] * fix_K * mu^(j - c1 fix_TQ + fix_HhQ)       */
    const void *in_ring; uint64_t in_mask; int64_t in_abs0;
    const void *fixV; const float *fix_T1, *fix_T2; unsigned fix_TQ, fix_HhQ; float fix_K;
    const float *fix_G; float fix_rTQ;   /* k_fe_level2: fix_G[e] = fix_K * (fix_T1[e >> 5] * fix_T2[e & 31]) as one table; 1 / fix_TQ */
    unsigned fix_limit;         /* new samples with index >= fix_limit were already corrected in place (k_fe_carry)   */
    uint32_t step_rinv;         /* floor(2^56 / step) clamped to 32 bits: integer ceil-division by the resampler step */
    void *tile_j;               /* nullable [ntiles][2] u64: the tile's resampler output range [ja, jb), for k_fe_tilefix */
    int m[PMR_FE_MAX_STAGES], tap_off[PMR_FE_MAX_STAGES];
    float dc_a1, zeta, lam_wave;
    float lam_pow16[6];         /* lambda^(spt * 2^j)                                                */
    int taps_valid;             /* taps_k holds all taps of this launch (they fit)                    */
    float taps_k[64];           /* copy of this launch's branch taps (its stages only, execution order) in the kernel
                                   argument segment: the specialised kernel reads them as scalars, no pointer chase */
} pmr_fe_params;

typedef struct {
    const void *probeA, *probeB, *probeL, *probeE, *v_in;
    void *v_out, *V;
    unsigned ntiles, K, c_end;
    float rho, lamHh, inv_lamHh, inv_lamL, lamEnd;
    const float *rho_pow;       /* [K + 1] rho^k (k_fe_tilefix)                                       */
    const void *tile_j;         /* [ntiles][2] u64: resampler output range of every tile, published by k_frontend* */
} pmr_fe_tiles_params;

typedef struct {
    void *xr; uint64_t pos0, mask; const void *V;
    const float *GA, *T1, *T2;  /* one-level form (k_fe_tilefix, k_fe_carry_tail): GA = Kgain * gain per branch; level 1 (k_fe_carry): unused */
    unsigned ny, TQ, HhQ;       /* ny: outputs of the block; only j in [j0, ny) are corrected (k_fe_carry, k_fe_tilefix) */
    unsigned j0;
    uint32_t phi0, step;
    float Kgain;
} pmr_fe_fix_params;

/* A launch can carry its own events (hipExtLaunchKernel): `stop` is the dispatch packet's completion signal -- no packet of its
 * own on the stream, where hipEventRecord is a marker packet that costs ~3.6 us between two back-to-back kernels -- and
 * `start` a marker in front (start..stop = the kernel's own begin..end timestamps).  Both nullable hipEvent_t; ev may be NULL. */
typedef struct { void *start, *stop; } pmr_launch_events;
int pmr_launch_frontend(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, int nt, int spt,
                        const pmr_launch_events *ev);
/* level 2 of the two-level front end: the specialised k_fe_level2 (m = 5, 10 + resampler, 2048-sample tiles) or the generic
 * k_frontend in mode 2 (4096-sample tiles) */
int pmr_launch_frontend_l2(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, int fast);
/* specialised kernels of pmr_fe_fast.hip; return -1 when the cascade is not one they cover */
int pmr_launch_fe_fast(pmr_stream_t s, const pmr_fe_params *p, unsigned ntiles, const pmr_launch_events *ev);
/* does a specialised kernel exist for this cascade (mode 0: whole front end, 1: level 1, 2: level 2; m[0..h): stages in execution
 * order)?  Six-tap stages followed by one of the (MA, MB) pairs pmr_fe_fast.hip instantiates */
int pmr_fe_fast_covers(int mode, const int *m, int h);
/* tile carries of a level-1 launch (V[c], next call's dc state) + in-place dc fix of the ring samples [f->j0, f->ny) */
int pmr_launch_fe_carry(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f);
/* one-level front end, carry applied at the channelizer's loads: tile carries + in-place correction of the block's tail [f->j0, f->ny) */
int pmr_launch_fe_carry_tail(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f, const pmr_launch_events *ev);
/* one-level front end, carries + correction of the resampled stream in place, one wave per tile */
int pmr_launch_fe_tilefix(pmr_stream_t s, const pmr_fe_tiles_params *t, const pmr_fe_fix_params *f, unsigned n_q,
                          const pmr_launch_events *ev);

/* waterfall periodogram (pmr_spectrum.hip): PSD (linear, averaged, fft-shifted, 4 wlen bins) of ny ring samples from pos0 */
unsigned pmr_spgram_max_workgroups(void);
int pmr_launch_spgram(pmr_stream_t s, const void *xr, uint64_t xr_mask, uint64_t pos0, unsigned ny, unsigned wlen,
                      const float *win, const void *tw, float *partial /*[max workgroups][4 wlen]*/, float *psd_mag /*[4 wlen]*/);

/* TEST-ONLY poison mode (pmr_poison.hip; pmr_debug_poison in include/pmr_chain.h, or PMR_DEBUG_POISON=1 in the environment):
 * every launch of this library is preceded by a kernel that overwrites ALL LDS of EVERY CU with a signalling-NaN pattern, and the
 * scratch buffers of pmr_chain.c are filled with 0xFF bytes instead of zeros -- a kernel whose result depends on bytes it did not
 * write then fails on every box, not on the one box where the stale bytes happen to be a NaN. */
int pmr_debug_poison_enabled(void);
int pmr_debug_poison_lds(pmr_stream_t s);
/* 1 when the KERNEL units were compiled with -DPMR_EXPERIMENT (pmr_experiment.h); pmr_chain_info ORs it with the host units' own */
int pmr_kernels_experiment_build(void);

#ifdef __cplusplus
}
#endif

#if defined(__HIPCC__)
/* timing experiment (tools/ab_libs.py, -DEXP_BE_WIN): the back end's global STORES land in small windows (discriminator rows & 63, PCM
 * positions & 1023): what do the back end's write streams cost the chain?  WRONG results */
#ifdef EXP_BE_WIN
#define PMR_EXP_ROW_AND 63ull
#define PMR_EXP_PCM_AND 1023l
#else
#define PMR_EXP_ROW_AND (~0ull)
#define PMR_EXP_PCM_AND (~0l)
#endif
/* every kernel launch of the library goes through here (poison mode above; otherwise exactly hipLaunchKernelGGL) */
#define PMR_KLAUNCH(kern, grid, block, lds, st, ...)                                                                               \
    do {                                                                                                                           \
        if (pmr_debug_poison_enabled()) (void)pmr_debug_poison_lds((pmr_stream_t)(st));                                            \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                               \
    } while (0)

/* arg(re + j im) for the discriminator (freqdem, reference src/sdr_pmr446.c:881: cargf(conj(r') r)): the library atan2f spends a
 * third of its ~38 instructions on range scaling (frexp / ldexp of both operands) and on inf / NaN classes.  Channelizer outputs
 * are ordinary floats, so: q = min / max by one v_rcp_f32, the same odd minimax polynomial the library evaluates, octant and
 * sign fix-ups.  Differs from atan2f in the last bit at most (|d| <= 2.4e-7 rad ~ 1e-7 of full scale after the 1 / pi);
 * the signs of zeros are honoured like atan2f's (arg(+0 + j0) = 0, arg(-0 + j0) = pi: what liquid's freqdem yields on its first
 * sample, whose conj(0) r product has a negative-zero real part for re(r) > 0). */
static __device__ __forceinline__ float pmr_arg(float im, float re)
{
#ifdef EXP_ARG_CHEAP        /* timing experiment: what do the discriminator's ~24 instructions per channel and frame cost the chain?  WRONG results */
    return im * re;
#endif
    const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    float q = mn * __builtin_amdgcn_rcpf(mx);
    q = mx == 0.0f ? 0.0f : q;
    const float s = q * q;
    float p = __builtin_fmaf(s, 0x1.5a54bp-9f, -0x1.f4b218p-7f);
    p = __builtin_fmaf(s, p, 0x1.53f67ep-5f);
    p = __builtin_fmaf(s, p, -0x1.2fa9aep-4f);
    p = __builtin_fmaf(s, p, 0x1.b26364p-4f);
    p = __builtin_fmaf(s, p, -0x1.22c1ccp-3f);
    p = __builtin_fmaf(s, p, 0x1.99717ep-3f);
    p = __builtin_fmaf(s, p, -0x1.5554c4p-2f);
    float r = __builtin_fmaf(q, s * p, q);
    r = ay > ax ? 0x1.921fb6p+0f - r : r;
    r = __builtin_bit_cast(int, re) < 0 ? 0x1.921fb6p+1f - r : r;      // the SIGN BIT: atan2(+-0, -0) = +-pi, as cargf gives for conj(0) r
    return __builtin_copysignf(r, im);
}

#include <atomic>
/* hipFuncSetAttribute is per device, and a process may hold handles on several GPUs driven from several threads
 * (pmr_chain_cfg.device): the "dynamic-LDS limit already raised" flag is one bit per device ordinal in an ATOMIC word.
 * Two threads racing on the same bit both raise the limit -- harmless; none ever skips it. */
typedef std::atomic<unsigned long long> pmr_attr_flags;
static inline bool pmr_attr_needed(pmr_attr_flags &mask)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    return !(mask.fetch_or(1ull << dev, std::memory_order_relaxed) >> dev & 1ull);
}
/* XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2), so workgroup b takes LOGICAL
 * index (b % 8) * (n / 8) + b / 8 -- every XCD then works through a CONTIGUOUS range of tiles, and the rows two neighbouring
 * tiles both read (filter history: 25 of 33 rows in k_pfb_wide, 408 of 664 in the FIR window) are fetched into one L2 once
 * instead of into two L2s.  Placement only: never changes results. */
static __device__ __forceinline__ unsigned pmr_xcd_contiguous(unsigned b, unsigned n)
{
    const unsigned per = n >> 3, main = per << 3;
    return b < main ? (b & 7u) * per + (b >> 3) : b;
}
#endif
#endif
