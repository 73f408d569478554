/* pmr_chain.h -- C-ABI of the MI355X-native per-block IQ DSP chain (libpmr446_hip.so).
 *
 * One call = one iteration of the reference's block loop body, src/sdr_pmr446.c:795-906:
 *   dc-block (:795) -> msresamp (:796) -> ring carry (:797,:804-805,:815) -> NCO shift (:808-812)
 *   -> firpfbch analyzer (:814) -> transpose (:819-821) -> freqdem (:881) -> CTCSS high-pass (:882)
 *   -> gain (:890) -> de-emphasis (:895-899) -> optional low-pass (:900-902) -> audio hand-off (:903-906)
 * executed by hand-written gfx950 HIP kernels, for ALL M channels (the reference demodulates the one
 * squelch-selected channel, :876-877).  The reference has no process-one-block function (the body is inline
 * in main()); this header defines it, keeping liquid-dsp's conventions at the call sites it replaces:
 * opaque handle, xxx_create() returns NULL on failure, int return codes with 0 == OK (LIQUID_OK), the
 * caller owns every sample buffer, one thread per handle, handles are independent (one IQ stream = one
 * GPU = one handle; that is the multi-GPU model, no collectives).
 *
 * Plain C: no C++ or torch types.  Binding sketch for the reference's main(): see INTEGRATION.md.
 */
#ifndef PMR_CHAIN_H
#define PMR_CHAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
typedef struct { float re, im; } pmr_cf32;          /* layout-identical to C99 float _Complex */
#else
typedef float _Complex pmr_cf32;                    /* == SOAPY_SDR_CF32 sample, src/shared.c:62 */
#endif

typedef struct pmr_chain_s *pmr_chain;

/* Runtime replacements for the #defines / literals of src/sdr_pmr446.c:18-46 and init_liquid() :420-480. */
typedef struct {
    double   fs_in;              /* include/sdr_pmr446.h:13  SDR_SAMPLERATE      1024000            */
    unsigned num_channels;       /* src/sdr_pmr446.c:23      NUM_CHANNELS        16 (power of two)  */
    double   channel_width_hz;   /* :22                      CHANNEL_WIDTH_HZ    12500              */
    float    dcblock_alpha;      /* :422  iirfilt_crcf_create_dc_blocker         0.0005f            */
    float    resamp_As;          /* :426  msresamp_crcf_create                   60.0f              */
    unsigned pfb_m;              /* :437  firpfbch_crcf_create_kaiser            13                 */
    float    pfb_As;             /* :437                                         80.0f              */
    float    fm_kf;              /* :440  freqdem_create                         0.5f               */
    float    audio_gain;         /* :33,:890  SDR_DEFAULT_AUDIO_GAIN             4.0f               */
    int      lowpass;            /* :154,:900  args.lowpass                      0                  */
    int      deemph_fir;         /* :457  APP_FIR_DEEMPH                         0 = IIR            */
    unsigned max_block;          /* :30   SDR_INPUT_CHUNK                        100000             */
    int      device;             /* HIP device ordinal; -1 = current device                         */
    /* Fixed-coefficient tables the app hands to firfilt_rrrf_create (:443,:453,:458).
     * NULL selects the PMR446 tables of src/sdr_pmr446.c:56-136.                                   */
    const float *hp_taps;     unsigned hp_len;      /* 377 */
    const float *lp_taps;     unsigned lp_len;      /* 103 */
    const float *deemph_taps; unsigned deemph_len;  /* 101 */
} pmr_chain_cfg;

/* error codes (0 == OK, like LIQUID_OK) */
enum {
    PMR_OK = 0,
    PMR_EINVAL = 1,      /* bad argument / configuration                  */
    PMR_ERANGE = 2,      /* n_in > max_block, or stride < frames produced */
    PMR_EHIP = 3,        /* a HIP runtime call or kernel launch failed.  ANY failure inside a block (after the plan checks, which
                            return PMR_EINVAL / PMR_ERANGE with nothing advanced) leaves the stream position undefined: the handle
                            then refuses further blocks (PMR_EHIP) until pmr_chain_reset */
    PMR_ENOMEM = 4
};

void      pmr_chain_default_cfg(pmr_chain_cfg *cfg);       /* the reference's operating point            */
pmr_chain pmr_chain_create(const pmr_chain_cfg *cfg);      /* NULL on failure (incl. no HIP device)      */
int       pmr_chain_reset(pmr_chain q);                    /* all carried state to zero (stream restart) */
int       pmr_chain_destroy(pmr_chain q);
/* Why the last pmr_chain_create of THIS thread returned NULL ("" after a success): no handle exists to ask. */
const char *pmr_chain_create_error(void);
/* Restart the stream AT sample index n_raw: all carried state zero, counters (resampler phase, ring / NCO / frame positions, the
 * CTCSS detector's block grid) as after n_raw zero samples -- the reference's loop never ends (:788) and its counters pass 2^32
 * within seconds at the larger configurations.
 *  - It IS a pmr_chain_reset followed by setting the counters: blocks submitted but not collected (pmr_chain_submit_block) and
 *    blocks still in flight on the device are DROPPED, not drained -- collect first what you want to keep; a faulted handle
 *    (PMR_EHIP after a block failed mid-way) is usable again afterwards, exactly as after pmr_chain_reset.
 *  - n_raw >= 2^62: PMR_ERANGE, the handle is left untouched (position and state as before the call).
 * pmr_chain_position reports where the stream stands (any pointer may be NULL): the counters are the HOST's, advanced when a block is
 * ENQUEUED -- after un-synchronised device calls (pmr_chain_process_block_device) it is the position the queued work will reach,
 * not what the device has completed; pmr_chain_synchronize first where that matters. */
int       pmr_chain_seek(pmr_chain q, uint64_t n_raw);
void      pmr_chain_position(pmr_chain q, uint64_t *n_raw, uint64_t *n_resampled, uint64_t *n_frames);
unsigned  pmr_chain_max_frames(pmr_chain q);               /* SDR_CHANNEL_BUF_SIZE rule, :730-736        */
unsigned  pmr_chain_num_channels(pmr_chain q);
const char *pmr_chain_last_error(pmr_chain q);

/* Process one block held in HOST memory (== buffp after readStream, :789).
 *   iq        [n_in] cf32 interleaved, n_in <= max_block (0 allowed)
 *   pcm       [M][pcm_stride] int16, channel-major like ch_buff_mat_t (:51); nullable
 *   n_frames  frames produced this call (== ns, :800-823); nullable
 *   chan_out  [M][pcm_stride] cf32 channelizer tap-off (== chan_bufs, :743) for squelch/RSSI plumbing; nullable
 *   rssi_db   [M] 20*log10(mean|x|) per channel (== average_power, :330-336); nullable
 * PCM rule: (int16_t)(x * 32767) truncated toward zero (src/dsd_in.c:174), saturated.
 * Synchronous: returns after the results are in the caller's buffers.                                */
int pmr_chain_process_block(pmr_chain q, const pmr_cf32 *iq, unsigned n_in,
                            int16_t *pcm, unsigned pcm_stride, unsigned *n_frames,
                            pmr_cf32 *chan_out, float *rssi_db);

/* Same, plus the float32 audio the reference hands to its sink (:904), [M][pcm_stride]; nullable.    */
int pmr_chain_process_block_f32(pmr_chain q, const pmr_cf32 *iq, unsigned n_in,
                                int16_t *pcm, float *audio, unsigned pcm_stride, unsigned *n_frames,
                                pmr_cf32 *chan_out, float *rssi_db);

/* Same call on the receiver's OWN sample format (include/pmr_io.h: iq_format 0 = cf32, 1 = interleaved int16 / 32768, 2 = interleaved
 * uint8, (x - 127.5) / 127.5 -- the reference's radio is an RTL-SDR, README.md:12, whose uint8 pairs SoapySDR widens to the cf32 that
 * readStream hands over, src/shared.c:62, src/sdr_pmr446.c:789).  The samples are converted on the device, by the front end as it
 * loads them: a block in pmr_host_alloc memory of up to 2^18 samples is read IN PLACE, 2 or 4 bytes per sample instead of 8 on the
 * host link (100 000 uint8 samples: 200 KB instead of 800 KB per call).  PCM / audio are bit-identical to
 * pmr_chain_process_block_f32 on the converted samples. */
int pmr_chain_process_block_fmt(pmr_chain q, const void *iq, int iq_format, unsigned n_in,
                                int16_t *pcm, float *audio, unsigned pcm_stride, unsigned *n_frames,
                                pmr_cf32 *chan_out, float *rssi_db);

/* Two-step synchronous form, for a squelch that decides on THIS block before it is demodulated -- the reference's order: state
 * machine on chan_bufs (:828-874), then freqdem .. audio of the active channel (:876-906).  pmr_chain_channelize_block runs the
 * block up to channelizer / discriminator / RSSI (chan_out, rssi_db: as in pmr_chain_process_block); the caller updates
 * pmr_chain_set_channel_mask / pmr_chain_reset_channel; pmr_chain_demodulate_block then runs the audio part (high-pass ..
 * PCM, CTCSS branch) of that same block for the channels open NOW.  With an unchanged mask the pair returns exactly what
 * pmr_chain_process_block_f32 returns.  A channelized block that is never demodulated is still pushed through the stateful
 * audio stages (CTCSS detector, follow-on FIR passes) by the next call. */
int   pmr_chain_channelize_block(pmr_chain q, const pmr_cf32 *iq, unsigned n_in, unsigned *n_frames,
                                 pmr_cf32 *chan_out /*nullable [M][chan_stride]*/, unsigned chan_stride, float *rssi_db /*nullable [M]*/);
int   pmr_chain_demodulate_block(pmr_chain q, int16_t *pcm /*nullable*/, float *audio /*nullable*/, unsigned pcm_stride,
                                 unsigned *n_frames);

/* Asynchronous host-buffer pair: the call pattern of the reference's loop (one readStream block per iteration,
 * src/sdr_pmr446.c:789-796) with the sink one or two blocks behind.  submit queues H2D copy -> chain -> D2H copy of one block
 * and returns at once; collect waits for the OLDEST submitted block and hands over its outputs (arguments as for
 * pmr_chain_process_block_f32; outputs that were not requested in `want` are left untouched).  Up to
 * pmr_chain_max_in_flight() blocks may sit between the two, so the transfers of one block overlap the kernels of its
 * neighbours.  `iq` must stay valid and unchanged until the block has been collected; the copies are truly asynchronous only
 * from pinned memory -- pmr_host_alloc() returns memory pinned in THIS library's HIP runtime. */
enum { PMR_WANT_PCM = 1, PMR_WANT_AUDIO = 2, PMR_WANT_RSSI = 4, PMR_WANT_CHAN = 8 };
int      pmr_chain_submit_block(pmr_chain q, const pmr_cf32 *iq, unsigned n_in, unsigned want /*PMR_WANT_*; 0 = PCM*/);
/* the same for the other ingest formats of include/pmr_io.h (iq_format: 0 = cf32, 1 = interleaved int16 / 32768, 2 = interleaved
 * uint8 as (x - 127.5) / 127.5, the rtl_sdr format): the samples cross PCIe as they come from the receiver and are converted
 * on the device -- 2 or 4 bytes per sample on the host link instead of 8 */
int      pmr_chain_submit_block_fmt(pmr_chain q, const void *iq, int iq_format, unsigned n_in, unsigned want);
int      pmr_chain_collect_block(pmr_chain q, int16_t *pcm, float *audio, unsigned pcm_stride, unsigned *n_frames,
                                 pmr_cf32 *chan_out, float *rssi_db);
unsigned pmr_chain_blocks_in_flight(pmr_chain q);
unsigned pmr_chain_max_in_flight(pmr_chain q);
/* Pinned, device-visible host memory.  Besides making submit's copies asynchronous it lets the SYNCHRONOUS entry points skip
 * the copy engines for small blocks: an `iq` of up to 2^18 samples that lies inside a pmr_host_alloc allocation is read by the
 * front end in place over the host link, and the outputs are written straight into pinned staging (a 100 000-sample call:
 * 93 us instead of 103 us; larger blocks and other memory go through H2D / D2H copies as before). */
void    *pmr_host_alloc(size_t bytes);                     /* NULL on failure */
void     pmr_host_free(void *p);

/* Device-resident variant: every pointer is a HIP device pointer on the chain's device; the call does not wait for the block
 * (call pmr_chain_synchronize).  Back-pressure: at most pmr_chain_max_in_flight() blocks are ever queued -- a call made while
 * that many are in flight first waits (on the host) for the oldest one's back end, whose ring space it reuses.
 * n_frames is a host pointer and is valid on return (frame counts are closed-form in n_in).
 * STREAM CONTRACT.  Outputs are written by work queued on pmr_chain_stream().  d_iq is read by the front end, which
 * pipelined calls queue on a SECOND, internal stream that does not wait for pmr_chain_stream() or the null stream:
 *   - d_iq must be complete when the call is made, OR the caller records an event behind its producer and hands it over with
 *     pmr_chain_wait_input_event() just before the call (the reads then wait for it on the device);
 *   - d_iq must stay untouched until the block's front end has run: pmr_chain_synchronize_input() (or
 *     pmr_chain_synchronize()) returns once that is true for every queued block.  Up to pmr_chain_max_in_flight() front
 *     ends may be queued ahead of their back ends.                                                                   */
int pmr_chain_process_block_device(pmr_chain q, const void *d_iq, unsigned n_in,
                                   void *d_pcm, void *d_audio, unsigned pcm_stride, unsigned *n_frames,
                                   void *d_chan_out, void *d_rssi_db);
int   pmr_chain_synchronize(pmr_chain q);
int   pmr_chain_wait_input_event(pmr_chain q, void *hip_event /* hipEvent_t */);
int   pmr_chain_synchronize_input(pmr_chain q);

/* ---- SURVEY s8 row f1, second half: demodulate only the OPEN channels (the reference's own semantics, src/sdr_pmr446.c:876-877;
 * the squelch state machine hands over the active channel, :834-839).  mask_words: bit (k & 63) of word k >> 6 enables
 * channel k, n_words * 64 >= M; NULL = every channel (the default).  Channelizer, RSSI and the discriminator keep running for
 * every channel, so a channel that is opened later starts with current filter history; the audio FIR / PCM / CTCSS branch
 * run for the enabled channels only and the pcm / audio rows of disabled channels are left untouched (host and device entry
 * points alike).  Needs num_channels to be a multiple of 16 (the audio kernels' tile width); PMR_EINVAL otherwise.
 * pmr_chain_reset_channel = freqdem_reset + ctcss_detector_reset of one channel (what the reference does when the squelch
 * detunes, :866-867): the channel's first discriminator output of the next block is arg(0) = 0. ---- */
int   pmr_chain_set_channel_mask(pmr_chain q, const uint64_t *mask_words, unsigned n_words);
int   pmr_chain_reset_channel(pmr_chain q, unsigned channel);
/* Consecutive blocks pipeline on two HIP streams (front end of block b+1 under the back end of block b); 0 runs
 * every block start-to-finish before the next one (kernel timings then are uncontended).  Default: on.          */
int   pmr_chain_set_overlap(pmr_chain q, int on);
void *pmr_chain_stream(pmr_chain q);                       /* hipStream_t the kernels are launched on   */

/* ---- measurement hooks (bench.py): HIP-event time of every kernel launched by this handle ---- */
/* mode 0 off; 1 every kernel (event records around each launch: ~7 us of marker packets per launch, for un-pipelined breakdown
 * passes); m >= 2 only the front-end kernel, every (m-1)-th launch, by start/stop events the launch itself carries
 * (hipExtLaunchKernel: the kernel's own begin..end, ~5 us per sampled launch and nothing on the others) */
int         pmr_chain_profile_enable(pmr_chain q, int mode);
int         pmr_chain_profile_reset(pmr_chain q);
unsigned    pmr_chain_profile_count(pmr_chain q);                          /* number of distinct kernels   */
const char *pmr_chain_profile_name(pmr_chain q, unsigned i);
int         pmr_chain_profile_get(pmr_chain q, unsigned i, double *total_ms, unsigned *launches);

/* ---- introspection (tests): designed coefficients / integers, and the last block's intermediates ---- */
enum { PMR_INFO_NUM_STAGES = 0, PMR_INFO_M_STAGE = 1, PMR_INFO_ARB_STEP = 2, PMR_INFO_NCO_DTHETA = 3,
       PMR_INFO_ARB_NPFB = 4, PMR_INFO_ARB_M = 5, PMR_INFO_PFB_P = 6,
       PMR_INFO_CARRY_AT_LOAD = 7 /* 1: the front end's dc carry is applied where the channelizer loads the resampled stream
                                     (one-level front ends with the 16- / 256-channel kernels); handle only */,
       /* which kernels this handle's plan selected (handle only; tests assert that every fallback is reachable by a legal cfg) */
       PMR_INFO_FE_PLAN = 8       /* 0 staged (one kernel per stage: cascades too deep for an LDS tile), 1 generic tile kernel,
                                     2 specialised one-level (k_fe_fast), 3 two levels, specialised, 4 two levels, generic    */,
       PMR_INFO_CHAN_PLAN = 9     /* 0 generic k_channelize, 1 16-channel k_channelize_win, 2 fused 256-channel kernel,
                                     3 wide bank: k_pfb_wide + k_fft_disc                                                    */,
       PMR_INFO_FIR_PLAN = 10     /* 0 k_fir_pair (VALU), 1 direct MFMA form only, 2 FFT form for large blocks + direct MFMA  */,
       PMR_INFO_EXPERIMENT_BUILD = 11 /* 1: this LIBRARY was compiled with -DPMR_EXPERIMENT, the one gate of every compile-time experiment
                                     hook (timing-only builds with wrong results, tile-shape knobs: csrc/pmr_experiment.h).  The product
                                     build returns 0.  The only query that needs no handle: pmr_chain_info(NULL, 11, 0)               */ };
enum { PMR_DESIGN_HALFBAND = 0, PMR_DESIGN_ARB = 1, PMR_DESIGN_PFB = 2,
       PMR_DESIGN_DEEMPH = 3 /* {b0, b1, a1} of the de-emphasis IIR, normalised by a0 (src/sdr_pmr446.c:462-463) */ };
unsigned pmr_chain_info(pmr_chain q, int what, unsigned idx);
unsigned pmr_chain_design(pmr_chain q, int what, unsigned idx, float *out, unsigned cap);
enum { PMR_DEBUG_RESAMPLED = 0,   /* cf32 [ny]  resampler output of the last block (:796)              */
       PMR_DEBUG_FM = 1,          /* f32 [ns][M] discriminator output of the last block, time-major     */
       PMR_DEBUG_CTCSS_LP = 2 };  /* f32 [ns][M] CTCSS low-pass branch delay188(x) - hp(x) (:889) of the last block (detector on) */
int pmr_chain_debug_enable(pmr_chain q, int on);   /* capture the intermediates of subsequent blocks */
int pmr_chain_debug_read(pmr_chain q, int what, void *host_buf, size_t cap_bytes, size_t *n_bytes);
/* TEST-ONLY poison mode, process-wide (also PMR_DEBUG_POISON=1 in the environment; returns the previous setting).  While on,
 * every kernel launch of the library is preceded by a kernel that overwrites ALL LDS of EVERY CU with a signalling-NaN pattern,
 * and handles created while it is on fill their scratch buffers with 0xFF bytes instead of zeros: a kernel whose result depends
 * on bytes it did not write fails on every box.  Results are unchanged by it (tests/conftest.py runs the -m gpu tier under it);
 * costs ~25 us per launch. */
int pmr_debug_poison(int on);
/* the checker's checker: n_wg workgroups copy the first `words` (<= 40960: the whole 160 KiB of a CU) 32-bit words of their UNINITIALISED dynamic LDS to
 * d_out[n_wg][words] (device pointer) and the call waits for them; with the poison mode on every word reads 0x7FA0DEAD */
int pmr_debug_lds_probe(void *d_out, unsigned words, unsigned n_wg);

/* ---- CTCSS tone detection for every channel (complementary low-pass branch src/sdr_pmr446.c:884-889,
 * ctcss_execute :605-628, 38-tone Goertzel bank over 2441-sample blocks :366-409).  When enabled, every
 * process_block call also runs the detector; each Goertzel block completed by the call yields one event per channel.  With a
 * channel mask set the detector runs for the OPEN channels only (the reference calls ctcss_execute for the active channel, :893):
 * events of closed channels read {index -1, detected 0} (closed WHEN THE BLOCK RAN: a later set_channel_mask does not change
 * what pmr_chain_ctcss_read returns for it), and a channel's partial Goertzel sums restart when it is opened or reset.
 * Deviation from the reference, documented: the 2441-frame block GRID is the stream's, shared by all M channels, where the
 * reference's single detector restarts its own sample count at ctcss_detector_reset (:348-357, :867).  The block in progress when
 * a channel is opened / reset in mid-block is therefore incomplete for it and reports {-1, 0, 0, 0} ("no decision"); its first
 * real event comes at the first grid boundary >= 2441 frames after the restart (reference: exactly 2441 frames after).  The
 * detector's dc blocker (:606) is never reset, as in the reference. ---- */
typedef struct { int index;        /* strongest of the 38 tones (ctcss_freqs[index], :138-141)          */
                 int detected;     /* avg power > 120 && max/avg > 10 (:403-404)                         */
                 float max_power, avg_power; } pmr_ctcss_event;
int pmr_chain_ctcss_enable(pmr_chain q, int on);
/* events of the LAST process_block call: events[k * cap + e], e < *n_events (same count for every channel).
 * Synchronises the chain's streams -- the detector of pipelined calls runs on an internal stream of its own, so the events
 * are complete after this call (or pmr_chain_synchronize), not merely after work ordered behind pmr_chain_stream(). */
int pmr_chain_ctcss_read(pmr_chain q, pmr_ctcss_event *events, unsigned cap, unsigned *n_events);
/* frequency in Hz of tone `index` (0..37: ctcss_freqs, src/sdr_pmr446.c:138-141; what :611 stores in chain->ctcss_freq), 0 outside */
float pmr_ctcss_freq(int index);

/* ---- the waterfall line -- asgramcf of the resampled stream (asgramcf_create(width) +
 * set_scale(-40, 2) src/sdr_pmr446.c:473-477; asgramcf_write(resamp_buf, ny) + asgramcf_execute per block :911-912).
 * When enabled every process_block call also averages the periodograms of ITS resampled samples (Hann window of `nfft` samples
 * every nfft / 2, 4 nfft-point transforms; execute resets liquid's spgram, so blocks are independent). ---- */
int pmr_chain_spectrum_enable(pmr_chain q, unsigned nfft /* display width: power of two 8..1024; 0 = off */);
/* PSD in dB of the LAST block: psd_db[4 nfft], bin i <-> frequency (i / (4 nfft) - 0.5) x the resampled rate.  *n_transforms =
 * periodograms averaged (0: the block had fewer than nfft / 2 resampled samples; psd_db zeroed).  Synchronises. */
int pmr_chain_spectrum_read(pmr_chain q, float *psd_db, unsigned cap, unsigned *n_transforms);
/* host: asgramcf_execute's peak search and character mapping; ascii[nfft + 1] (terminated) */
int pmr_asgram_ascii(const float *psd_db, unsigned nfft, unsigned n_transforms, float ref, float div, char *ascii, float *peakval,
                     float *peakfreq);

/* ---- channel select + squelch hysteresis on rssi_db (host logic; mirrors find_max_rssi_channel,
 * src/sdr_pmr446.c:668-700, and the proc_scanning / proc_tuned state machine, :828-874) ---- */
enum { PMR_SCANNING = 0, PMR_TUNED = 1 };
typedef struct { int state; int active_chan; float rssi; } pmr_squelch;
void pmr_squelch_init(pmr_squelch *s);
/* mask_words / n_words: the channels that take part, in the layout of pmr_chain_set_channel_mask (bit k & 63 of word k >> 6; the
 * reference's uint64_t channel_mask :18 generalised to any M); NULL = all.  n_words * 64 < M: -1 / no change. */
int  pmr_find_max_rssi_channel(const float *rssi_db, unsigned M, const uint64_t *mask_words, unsigned n_words, float *max_rssi);
int  pmr_squelch_update(pmr_squelch *s, const float *rssi_db, unsigned M, const uint64_t *mask_words, unsigned n_words,
                        float squelch_level /* :34 default 18 dB */, int lock_mode_max);

/* ---- host-only helpers: pure arithmetic, need no HIP device (used by the CPU test tier) ---- */
unsigned pmr_cfg_info(const pmr_chain_cfg *cfg, int what, unsigned idx);
unsigned pmr_cfg_design(const pmr_chain_cfg *cfg, int what, unsigned idx, float *out, unsigned cap);
unsigned pmr_cfg_max_frames(const pmr_chain_cfg *cfg);        /* :730-736 sizing rule                    */
/* The closed-form sample accounting process_block uses to size its launches: given the carried counters,
 * how many resampled samples (ny, :796) and frames (ns, :804-823) a block of n_in samples yields.      */
typedef struct { uint64_t n_raw; uint32_t arb_phase; unsigned leftover; } pmr_plan_state;
int pmr_cfg_plan_block(const pmr_chain_cfg *cfg, pmr_plan_state *st, unsigned n_in, unsigned *ny, unsigned *ns);

#ifdef __cplusplus
}
#endif
#endif
