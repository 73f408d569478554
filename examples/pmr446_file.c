/* pmr446_file -- headless stand-in for the reference's main() loops on a RECORDED cf32 stream:
 *
 *   pmr446_file chan <in.cf32|-> <out.wav> [fs_in] [num_channels] [channel|-1] [waterfall]
 *       the block loop of src/sdr_pmr446.c:788-908: read a chunk (:789) -> pmr_chain_process_block_f32 -> squelch state
 *       machine on the GPU's RSSI (:828-874) -> float32 WAV at 12.5 kHz.  channel >= 0 writes that channel (mono, like
 *       the RtAudio sink :585) and, like the reference, demodulates ONLY that channel (:876-877) and logs its CTCSS tone
 *       (ctcss_execute :605-628); -1 writes all channels (multi-channel WAV).  waterfall = W > 0 (a power of two): the
 *       reference's waterfall line per block (asgramcf of the resampled stream, :910-915) on stderr, W characters wide.
 *   pmr446_file scan <in.cf32|-> <out.wav> [fs_in] [num_channels]
 *       the reference's own behaviour end to end: the squelch state machine (:828-874) on the GPU's RSSI picks the active
 *       channel, ONLY that channel is demodulated (channel mask, :876-877), its discriminator / CTCSS state is reset when the
 *       squelch closes (:866-867), mono float32 WAV of whatever channel is open (silence is not written, like :903-906).  Like
 *       the reference it decides on a block's channelizer output and demodulates that same block: pmr_chain_channelize_block,
 *       squelch update, pmr_chain_demodulate_block.
 *   pmr446_file dsd <in.cf32|-> <out.s16|-> [fs_in]
 *       the loop of src/dsd_in.c:159-179: s16le mono 48 kHz, ready for `dsd -i -`.
 *
 * Everything numerical happens in libpmr446_hip.so; this file is plumbing. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pmr_chain.h"
#include "pmr_dsd.h"
#include "pmr_io.h"

static int run_chan(const char *in, const char *out, double fs, unsigned M, int only, unsigned waterfall)
{
    pmr_chain_cfg cfg;
    pmr_chain_default_cfg(&cfg);
    cfg.fs_in = fs; cfg.num_channels = M;
    pmr_chain q = pmr_chain_create(&cfg);
    if (!q) return 2;
    const unsigned S = pmr_chain_max_frames(q), C = only >= 0 ? 1 : M;
    pmr_iq_reader r = pmr_iq_reader_open(in, PMR_IQ_CF32);
    pmr_wav_writer w = pmr_wav_writer_open(out, PMR_WAV_F32, (unsigned)cfg.channel_width_hz, C);
    pmr_cf32 *iq = (pmr_cf32 *)malloc((size_t)cfg.max_block * sizeof(pmr_cf32));
    float *audio = (float *)malloc((size_t)M * S * sizeof(float)), *rssi = (float *)malloc(M * sizeof(float));
    int16_t *pcm = (int16_t *)malloc((size_t)M * S * sizeof(int16_t));
    if (!r || !w || !iq || !audio || !rssi || !pcm) { fprintf(stderr, "pmr446_file: cannot open / allocate\n"); return 3; }
    pmr_squelch sq;
    pmr_squelch_init(&sq);
    /* one channel selected: the reference's semantics -- only it is demodulated to audio, its CTCSS tone is tracked (:893) */
    pmr_ctcss_event *ct = NULL; unsigned ct_cap = 0; int ct_on = 0, ct_code = -1;
    if (only >= 0) {
        uint64_t mask[64] = {0};
        mask[(unsigned)only >> 6] = 1ull << ((unsigned)only & 63);
        ct_cap = S / 2441 + 2;
        ct = (pmr_ctcss_event *)malloc((size_t)M * ct_cap * sizeof(*ct));
        if (!ct || (unsigned)only >= M || M > 4096 || pmr_chain_set_channel_mask(q, mask, (M + 63) / 64) || pmr_chain_ctcss_enable(q, 1)) {
            fprintf(stderr, "pmr446_file: %s\n", pmr_chain_last_error(q)); return 3;
        }
    }
    float *psd = NULL; char *ascii = NULL;
    if (waterfall) {                                                                  /* :473-477 */
        psd = (float *)malloc((size_t)4 * waterfall * sizeof(float)); ascii = (char *)malloc(waterfall + 1);
        if (!psd || !ascii || pmr_chain_spectrum_enable(q, waterfall)) { fprintf(stderr, "pmr446_file: %s\n", pmr_chain_last_error(q)); return 3; }
    }
    int n, rc = 0;
    unsigned long blocks = 0, frames = 0;
    while ((n = pmr_iq_reader_read(r, iq, cfg.max_block)) > 0) {                      /* :789 */
        unsigned ns = 0;
        rc = pmr_chain_process_block_f32(q, iq, (unsigned)n, pcm, audio, S, &ns, NULL, rssi);
        if (rc) { fprintf(stderr, "pmr446_file: %s\n", pmr_chain_last_error(q)); break; }
        const int ev = pmr_squelch_update(&sq, rssi, M, NULL, 0, 18.0f, 0);               /* :828-874 */
        if (ev) fprintf(stderr, "block %lu: %s channel %d (%.1f dB)\n", blocks, sq.state == PMR_TUNED ? "tuned to" : "left",
                        sq.active_chan + 1, sq.rssi);
        rc = pmr_wav_writer_write_f32(w, only >= 0 ? audio + (size_t)only * S : audio, ns, S);
        if (rc) break;
        if (ct) {                                                                     /* the log lines of ctcss_execute, :613-626 */
            unsigned nev = 0;
            if ((rc = pmr_chain_ctcss_read(q, ct, ct_cap, &nev))) break;
            for (unsigned e = 0; e < nev; e++) {
                const pmr_ctcss_event *v = &ct[(size_t)only * ct_cap + e];
                if (v->index < 0) continue;                                               /* incomplete block after (re)opening: no decision */
                if (v->detected && !ct_on) fprintf(stderr, "Acquired CTCSS code: %d (frequency: %3.2fHz)\n", v->index + 1, pmr_ctcss_freq(v->index));
                else if (v->detected && v->index != ct_code) fprintf(stderr, "CTCSS code change: %d (frequency: %3.2fHz)\n", v->index + 1, pmr_ctcss_freq(v->index));
                else if (!v->detected && ct_on) fprintf(stderr, "Lost CTCSS code\n");
                ct_on = v->detected; ct_code = v->index;
            }
        }
        if (waterfall) {                                                              /* :910-915 */
            unsigned ntr = 0; float maxval = 0.f, maxfreq = 0.f;
            if ((rc = pmr_chain_spectrum_read(q, psd, 4 * waterfall, &ntr))) break;
            pmr_asgram_ascii(psd, waterfall, ntr, -40.0f, 2.0f, ascii, &maxval, &maxfreq);
            fprintf(stderr, " > %s < pk%5.1fdB [%5.2f] [max SNR: %5.1fdB]\n", ascii, maxval, maxfreq, sq.rssi);
        }
        blocks++; frames += ns;
    }
    fprintf(stderr, "pmr446_file: %lu blocks, %lu frames per channel\n", blocks, frames);
    pmr_wav_writer_close(w); pmr_iq_reader_close(r); pmr_chain_destroy(q);
    free(iq); free(audio); free(rssi); free(pcm); free(psd); free(ascii); free(ct);
    return rc || n < 0 ? 1 : 0;
}

static int run_scan(const char *in, const char *out, double fs, unsigned M)
{
    pmr_chain_cfg cfg;
    pmr_chain_default_cfg(&cfg);
    cfg.fs_in = fs; cfg.num_channels = M;
    pmr_chain q = pmr_chain_create(&cfg);
    if (!q || M > 4096) return 2;
    const unsigned S = pmr_chain_max_frames(q), W = (M + 63) / 64;
    pmr_iq_reader r = pmr_iq_reader_open(in, PMR_IQ_CF32);
    pmr_wav_writer w = pmr_wav_writer_open(out, PMR_WAV_F32, (unsigned)cfg.channel_width_hz, 1);
    pmr_cf32 *iq = (pmr_cf32 *)malloc((size_t)cfg.max_block * sizeof(pmr_cf32));
    float *audio = (float *)malloc((size_t)M * S * sizeof(float)), *rssi = (float *)malloc(M * sizeof(float));
    int16_t *pcm = (int16_t *)malloc((size_t)M * S * sizeof(int16_t));
    if (!r || !w || !iq || !audio || !rssi || !pcm) { fprintf(stderr, "pmr446_file: cannot open / allocate\n"); return 3; }
    uint64_t mask[64] = {0};
    int rc = pmr_chain_set_channel_mask(q, mask, W);                                    /* scanning: nothing is demodulated */
    pmr_squelch sq;
    pmr_squelch_init(&sq);
    int n = 0;
    unsigned long blocks = 0, frames = 0;
    while (!rc && (n = pmr_iq_reader_read(r, iq, cfg.max_block)) > 0) {                 /* :789 */
        unsigned ns = 0;
        rc = pmr_chain_channelize_block(q, iq, (unsigned)n, &ns, NULL, 0, rssi);        /* :795-823 + average_power */
        if (rc) { fprintf(stderr, "pmr446_file: %s\n", pmr_chain_last_error(q)); break; }
        const int was = sq.state == PMR_TUNED ? sq.active_chan : -1;
        if (pmr_squelch_update(&sq, rssi, M, NULL, 0, 18.0f, 0)) {                        /* :828-874 */
            memset(mask, 0, sizeof(mask));
            if (sq.state == PMR_TUNED) {
                mask[(unsigned)sq.active_chan >> 6] = 1ull << ((unsigned)sq.active_chan & 63);
                fprintf(stderr, "block %lu: tuned to channel %d (%.1f dB)\n", blocks, sq.active_chan + 1, sq.rssi);
            } else {
                fprintf(stderr, "block %lu: left channel %d\n", blocks, was + 1);
                if (was >= 0) rc = pmr_chain_reset_channel(q, (unsigned)was);          /* :866-867 */
            }
            if (!rc) rc = pmr_chain_set_channel_mask(q, mask, W);
        }
        if (!rc) rc = pmr_chain_demodulate_block(q, pcm, audio, S, &ns);                /* :876-902, the channel open NOW */
        if (rc) { fprintf(stderr, "pmr446_file: %s\n", pmr_chain_last_error(q)); break; }
        if (sq.state == PMR_TUNED) { rc = pmr_wav_writer_write_f32(w, audio + (size_t)sq.active_chan * S, ns, S); frames += ns; }   /* :903-906 */
        blocks++;
    }
    fprintf(stderr, "pmr446_file: %lu blocks, %lu audio frames written\n", blocks, frames);
    pmr_wav_writer_close(w); pmr_iq_reader_close(r); pmr_chain_destroy(q);
    free(iq); free(audio); free(rssi); free(pcm);
    return rc || n < 0 ? 1 : 0;                                                          /* n < 0: the reader failed */
}

static int run_dsd(const char *in, const char *out, double fs)
{
    pmr_dsd_cfg cfg;
    pmr_dsd_default_cfg(&cfg);
    cfg.fs_in = fs;
    pmr_dsd q = pmr_dsd_create(&cfg);
    if (!q) return 2;
    const unsigned cap = pmr_dsd_max_out(q);
    pmr_iq_reader r = pmr_iq_reader_open(in, PMR_IQ_CF32);
    pmr_wav_writer w = pmr_wav_writer_open(out, PMR_RAW_S16, (unsigned)cfg.audio_rate, 1);
    pmr_cf32 *iq = (pmr_cf32 *)malloc((size_t)cfg.max_block * sizeof(pmr_cf32));
    int16_t *pcm = (int16_t *)malloc((size_t)cap * sizeof(int16_t));
    if (!r || !w || !iq || !pcm) { fprintf(stderr, "pmr446_file: cannot open / allocate\n"); return 3; }
    int n, rc = 0;
    while ((n = pmr_iq_reader_read(r, iq, cfg.max_block)) > 0) {                      /* src/dsd_in.c:161 */
        unsigned nz = 0;
        rc = pmr_dsd_process_block(q, iq, (unsigned)n, pcm, NULL, cap, &nz);           /* :167-175 */
        if (rc) { fprintf(stderr, "pmr446_file: %s\n", pmr_dsd_last_error(q)); break; }
        if ((rc = pmr_wav_writer_write_s16(w, pcm, nz, cap))) break;                    /* :177-178 */
    }
    pmr_wav_writer_close(w); pmr_iq_reader_close(r); pmr_dsd_destroy(q);
    free(iq); free(pcm);
    return rc || n < 0 ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && !strcmp(argv[1], "chan"))
        return run_chan(argv[2], argv[3], argc > 4 ? atof(argv[4]) : 1024000.0, argc > 5 ? (unsigned)atoi(argv[5]) : 16,
                        argc > 6 ? atoi(argv[6]) : -1, argc > 7 ? (unsigned)atoi(argv[7]) : 0);
    if (argc >= 4 && !strcmp(argv[1], "scan"))
        return run_scan(argv[2], argv[3], argc > 4 ? atof(argv[4]) : 1024000.0, argc > 5 ? (unsigned)atoi(argv[5]) : 16);
    if (argc >= 4 && !strcmp(argv[1], "dsd"))
        return run_dsd(argv[2], argv[3], argc > 4 ? atof(argv[4]) : 1024000.0);
    fprintf(stderr, "usage: %s chan <in.cf32|-> <out.wav> [fs_in] [num_channels] [channel|-1] [waterfall]\n"
                    "       %s scan <in.cf32|-> <out.wav> [fs_in] [num_channels]\n"
                    "       %s dsd  <in.cf32|-> <out.s16|-> [fs_in]\n", argv[0], argv[0], argv[0]);
    return 64;
}
