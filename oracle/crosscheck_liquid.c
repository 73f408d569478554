/* crosscheck_liquid.c -- PIN for the oracle: runs the reference's hot path on REAL liquid-dsp v1.7.0 objects and diffs every
 * stage against the oracle's restatement (orc_chain) on the same synthetic IQ.  TEST INFRASTRUCTURE ONLY.
 *
 * The oracle is "parity unpinned" (oracle/README.md): liquid-dsp is not vendored under /root/reference, not installed in the
 * build image and not downloadable there.  This program is what turns that into "pinned" on any machine that has the library
 * (github.com/jgaeddert/liquid-dsp, tag v1.7.0, configured --enable-simdoverride like reference .github/workflows/build.yml:30-36):
 *
 *     make -C oracle crosscheck          # builds + runs when <liquid/liquid.h> is found, prints SKIP otherwise
 *
 * It drives exactly the liquid calls of the reference, in its order, with its parameters:
 *   object creation   src/sdr_pmr446.c:422-463  (dc blocker, msresamp 60 dB, VCO offset, firpfbch kaiser M/13/80 dB, freqdem 0.5,
 *                                                377-tap HP, delay 188, IIR de-emphasis)
 *   block loop body   :795-823 (dc-block, resample, ring, NCO mix-down, analyzer, transpose), :881-898 per channel
 *   PCM rule          src/dsd_in.c:174
 *   waterfall line    :473-477, :911-912 (asgramcf of the resampled stream) vs orc_asgramcf
 *   dsd_in            src/dsd_in.c:104,:170 (msresamp_rrrf 12.5 k -> 48 k) vs orc_msresamp_rrrf
 * for M = 16 channels at 1.024 MS/s (the reference's operating point) and, with arguments, any other (fs, M), demodulating
 * EVERY channel with its own set of per-channel objects (the oracle's generalisation) so all of them are compared.
 *
 * Output: per stage the largest |difference| (resampler output relative to its RMS, channelizer tap-off, discriminator, float
 * audio, int16 PCM in LSB).  Exit 0 when PCM agrees within 1 LSB on every channel that carries a signal and the float stages
 * within 1e-5 of their scale -- the tolerances the GPU is held to against the oracle (tests/test_gpu_parity.py).
 * A FAIL here means an Appendix-A assumption of SURVEY.md is wrong for v1.7.0 (candidates, in order of the survey's own
 * confidence marks: half-band output scaling, resamp_crcf phase width / filter-bank design, msresamp2 stage design constants);
 * the per-stage numbers say which one. */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <liquid/liquid.h>

#include "orc_chain.h"
#include "../sdr_pmr446_amd/data/pmr446_taps.h"

/* deterministic test signal: every other channel an FM tone, noise floor 30 dB down (plain LCG: no dependency on the python synth) */
static uint32_t lcg_state = 12345u;
static float lcg_uniform(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return ((lcg_state >> 8) + 0.5f) / 16777216.0f; }

static void make_signal(float complex *x, size_t n, double fs, unsigned M)
{
    const double amp = 0.5 / sqrt((double)M), sigma = amp * sqrt(fs / 12500.0 / 1000.0);
    for (size_t i = 0; i < n; i++) {
        const double t = (double)i / fs;
        double re = 0, im = 0;
        for (unsigned k = 0; k < M; k += 2) {
            const double fk = ((double)k - (M - 1) / 2.0) * 12500.0, fa = 400.0 + 37.0 * (k % 64);
            const double ph = 2 * M_PI * fk * t + 0.7 * k + (1500.0 / fa) * sin(2 * M_PI * fa * t);
            re += amp * cos(ph); im += amp * sin(ph);
        }
        const double r = sigma * sqrt(-log(lcg_uniform())), a = 2 * M_PI * lcg_uniform();
        x[i] = (float)(re + r * cos(a)) + I * (float)(im + r * sin(a));
    }
}

typedef struct { freqdem fm; firfilt_rrrf hp; iirfilt_rrrf de; } chan_objs;

int main(int argc, char **argv)
{
    const double fs = argc > 1 ? atof(argv[1]) : 1024000.0;
    const unsigned M = argc > 2 ? (unsigned)atoi(argv[2]) : 16;
    const unsigned block = 100000, nblocks = argc > 3 ? (unsigned)atoi(argv[3]) : 6;
    const float gain = 4.0f;
    const float rate = (float)(M * 12500.0) / (float)fs;
    printf("liquid-dsp %s vs oracle restatement: fs = %.0f, M = %u, %u blocks of %u samples\n", liquid_libversion(), fs, M, nblocks, block);

    /* ---- the reference's objects (:422-463) ---- */
    iirfilt_crcf dcblock = iirfilt_crcf_create_dc_blocker(0.0005f);
    msresamp_crcf resampler = msresamp_crcf_create(rate, 60.0f);
    nco_crcf nco = nco_crcf_create(LIQUID_VCO);
    nco_crcf_set_frequency(nco, -0.5f * (float)(M - 1) / (float)M * 2 * M_PI);
    firpfbch_crcf channelizer = firpfbch_crcf_create_kaiser(LIQUID_ANALYZER, M, 13, 80.0f);
    chan_objs *ch = calloc(M, sizeof(*ch));
    for (unsigned i = 0; i < M; i++) {
        ch[i].fm = freqdem_create(0.5f);
        ch[i].hp = firfilt_rrrf_create((float *)pmr446_hp_audio_taps, PMR446_HP_AUDIO_TAPS_LEN);
        float b[2] = {0.507301437230636f, 0.507301437230636f}, a[2] = {1.0f, 0.014602874461272194f};
        ch[i].de = iirfilt_rrrf_create(b, 2, a, 2);
    }
    const unsigned res_size = (unsigned)ceilf(1 + 2 * (float)block * rate), chan_size = res_size / M + 1;
    cbuffercf ring = cbuffercf_create(res_size + M);

    /* ---- the oracle on the same configuration ---- */
    orc_chain_cfg oc;
    orc_chain_default_cfg(&oc);
    oc.fs_in = fs; oc.num_channels = M; oc.max_block = block; oc.audio_gain = gain;
    orc_chain *o = orc_chain_create(&oc);
    if (!dcblock || !resampler || !nco || !channelizer || !ring || !o) { fprintf(stderr, "object creation failed\n"); return 2; }
    const unsigned S = orc_chain_max_frames(o), RS = orc_chain_max_resampled(o);

    float complex *x = malloc(sizeof(*x) * block), *xl = malloc(sizeof(*xl) * block), *res = malloc(sizeof(*res) * (res_size + 16));
    float complex *frame_out = malloc(sizeof(*frame_out) * M), *chan = malloc(sizeof(*chan) * (size_t)M * chan_size);
    float *t1 = malloc(sizeof(float) * chan_size), *t2 = malloc(sizeof(float) * chan_size);
    int16_t *pcm_o = malloc(sizeof(int16_t) * (size_t)M * S);
    float complex *chan_o = malloc(sizeof(*chan_o) * (size_t)M * S), *res_o = malloc(sizeof(*res_o) * RS);
    float *fm_o = malloc(sizeof(float) * (size_t)M * S), *au_o = malloc(sizeof(float) * (size_t)M * S);

    double d_res = 0, s_res = 0, d_chan = 0, s_chan = 0, d_fm = 0, d_audio = 0;
    int d_pcm = 0, count_mismatch = 0;
    for (unsigned b = 0; b < nblocks; b++) {
        make_signal(x, block, fs, M);
        memcpy(xl, x, sizeof(*x) * block);
        /* reference loop body, :795-823 */
        unsigned ny = 0, ns = 0;
        iirfilt_crcf_execute_block(dcblock, xl, block, xl);
        msresamp_crcf_execute(resampler, xl, block, res, &ny);
        cbuffercf_write(ring, res, ny);
        while (cbuffercf_size(ring) >= M) {
            float complex *rp; unsigned nr;
            cbuffercf_read(ring, M, &rp, &nr);
            for (unsigned i = 0; i < M; i++) { nco_crcf_mix_down(nco, rp[i], &rp[i]); nco_crcf_step(nco); }
            firpfbch_crcf_analyzer_execute(channelizer, rp, frame_out);
            cbuffercf_release(ring, nr);
            for (unsigned i = 0; i < M; i++) chan[(size_t)i * chan_size + ns] = frame_out[i];
            ns++;
        }
        /* oracle, same block */
        orc_taps taps; memset(&taps, 0, sizeof(taps));
        taps.resampled = (void *)res_o; taps.resampled_cap = RS; taps.fm = fm_o; taps.audio = au_o; taps.stride = S;
        unsigned ns_o = 0;
        if (orc_chain_process_block(o, (const void *)x, block, pcm_o, S, &ns_o, (void *)chan_o, NULL, &taps)) { fprintf(stderr, "oracle failed\n"); return 2; }
        if (ns_o != ns || taps.n_resampled != ny) { count_mismatch++; printf("block %u: counts differ: liquid ny=%u ns=%u, oracle ny=%u ns=%u\n", b, ny, ns, taps.n_resampled, ns_o); continue; }
        for (unsigned i = 0; i < ny; i++) { d_res = fmax(d_res, cabsf(res[i] - res_o[i])); s_res = fmax(s_res, cabsf(res[i])); }
        /* per channel, :881-898 */
        for (unsigned i = 0; i < M; i++) {
            const float complex *row = chan + (size_t)i * chan_size;
            for (unsigned k = 0; k < ns; k++) { d_chan = fmax(d_chan, cabsf(row[k] - chan_o[(size_t)i * S + k])); s_chan = fmax(s_chan, cabsf(row[k])); }
            freqdem_demodulate_block(ch[i].fm, (float complex *)row, ns, t1);
            firfilt_rrrf_execute_block(ch[i].hp, t1, ns, t2);
            for (unsigned k = 0; k < ns; k++) t2[k] *= gain;
            iirfilt_rrrf_execute_block(ch[i].de, t2, ns, t2);
            const int signal = (i % 2) == 0;                          /* odd channels carry noise only: discriminator ill-conditioned */
            for (unsigned k = 0; k < ns; k++) {
                if (!signal || (b == 0 && k < 700)) continue;        /* start-up: |chan| ~ 0, arg() ill-conditioned, rings through the FIR */
                d_fm = fmax(d_fm, fabsf(t1[k] - fm_o[(size_t)i * S + k]));
                d_audio = fmax(d_audio, fabsf(t2[k] - au_o[(size_t)i * S + k]));
                float s = t2[k] * 32767.0f;
                int q = s >= 32767.0f ? 32767 : s <= -32768.0f ? -32768 : (int)s;      /* src/dsd_in.c:174 + saturation */
                int dd = abs(q - (int)pcm_o[(size_t)i * S + k]);
                if (dd > d_pcm) d_pcm = dd;
            }
        }
    }
    /* ---- the waterfall line (:473-477 asgramcf_create + set_scale(-40, 2); :911-912 write(resamp_buf, ny) + execute): the last
     * block's resampled stream through liquid's asgramcf and the oracle's restatement; `res` still holds liquid's samples ---- */
    int ascii_diff = 0; double d_psd_peak = 0;
    {
        const unsigned W = 64;
        asgramcf ag = asgramcf_create(W);
        asgramcf_set_scale(ag, -40.0f, 2.0f);
        orc_asgramcf *og = orc_asgramcf_create(W);
        orc_asgramcf_set_scale(og, -40.0f, 2.0f);
        char a_l[W + 1], a_o[W + 1];
        float pv_l = 0, pf_l = 0, pv_o = 0, pf_o = 0;
        unsigned ny_last = 0;
        {   /* one more block through liquid's front end and the oracle's, to have both resampled streams side by side */
            make_signal(x, block, fs, M);
            memcpy(xl, x, sizeof(*x) * block);
            iirfilt_crcf_execute_block(dcblock, xl, block, xl);
            msresamp_crcf_execute(resampler, xl, block, res, &ny_last);
            orc_taps taps; memset(&taps, 0, sizeof(taps));
            taps.resampled = (void *)res_o; taps.resampled_cap = RS; taps.stride = S;
            unsigned ns_o = 0;
            orc_chain_process_block(o, (const void *)x, block, pcm_o, S, &ns_o, NULL, NULL, &taps);
        }
        asgramcf_write(ag, res, ny_last);
        asgramcf_execute(ag, a_l, &pv_l, &pf_l);
        orc_asgramcf_write(og, (const cf32 *)res_o, ny_last);
        orc_asgramcf_execute(og, a_o, &pv_o, &pf_o, NULL);
        a_l[W] = a_o[W] = 0;
        for (unsigned i = 0; i < W; i++) ascii_diff += a_l[i] != a_o[i];
        d_psd_peak = fabs((double)pv_l - (double)pv_o);
        printf("asgramcf (W = %u)  %d characters differ, peak %.2f dB vs %.2f dB at %.4f vs %.4f\n", W, ascii_diff, pv_l, pv_o, pf_l, pf_o);
        asgramcf_destroy(ag); orc_asgramcf_destroy(og);
    }

    /* ---- `dsd_in`'s interpolator (src/dsd_in.c:104,:170): msresamp_rrrf 12.5 kS/s -> 48 kS/s, As = 60 dB, on a discriminator-like
     * real stream, vs the oracle's orc_msresamp_rrrf; then the int16 rule of :172-175 ---- */
    double d_up = 0; int d_up_pcm = 0, up_count_mismatch = 0;
    {
        const float r_up = 48000.0f / 12500.0f;
        msresamp_rrrf up = msresamp_rrrf_create(r_up, 60.0f);
        orc_msresamp_rrrf *oup = orc_msresamp_rrrf_create(r_up, 60.0f);
        const unsigned n = 12500, cap = (unsigned)(n * r_up) + 64;
        float *u = malloc(sizeof(float) * n), *y_l = malloc(sizeof(float) * cap), *y_o = malloc(sizeof(float) * cap);
        for (unsigned b = 0; b < 3; b++) {
            for (unsigned i = 0; i < n; i++) {
                const double t = (double)(b * n + i) / 12500.0;
                u[i] = (float)(0.4 * sin(2 * M_PI * 1000.0 * t) + 0.2 * sin(2 * M_PI * 2417.0 * t + 1.0) + 0.01 * (lcg_uniform() - 0.5));
            }
            unsigned nl = 0, no = 0;
            msresamp_rrrf_execute(up, u, n, y_l, &nl);
            orc_msresamp_rrrf_execute(oup, u, n, y_o, &no);
            if (nl != no) { up_count_mismatch++; printf("interpolator block %u: liquid %u outputs, oracle %u\n", b, nl, no); continue; }
            for (unsigned i = 0; i < nl; i++) {
                d_up = fmax(d_up, fabsf(y_l[i] - y_o[i]));
                const int dd = abs((int)(int16_t)(y_l[i] * INT16_MAX) - (int)(int16_t)(y_o[i] * INT16_MAX));     /* :174 */
                if (dd > d_up_pcm) d_up_pcm = dd;
            }
        }
        printf("msresamp_rrrf x3.84  max |diff| %.3g, int16 %d LSB\n", d_up, d_up_pcm);
        msresamp_rrrf_destroy(up); orc_msresamp_rrrf_destroy(oup);
        free(u); free(y_l); free(y_o);
    }

    printf("resampler output  max |diff| %.3g  (scale %.3g)\n", d_res, s_res);
    printf("channelizer       max |diff| %.3g  (scale %.3g)\n", d_chan, s_chan);
    printf("discriminator     max |diff| %.3g\n", d_fm);
    printf("float audio       max |diff| %.3g\n", d_audio);
    printf("int16 PCM         max |diff| %d LSB\n", d_pcm);
    const int ok = !count_mismatch && d_pcm <= 1 && d_res <= 1e-5 * s_res && d_chan <= 1e-5 * s_chan &&
                   !up_count_mismatch && d_up <= 1e-5 && d_up_pcm <= 1 && ascii_diff == 0 && d_psd_peak <= 0.02;
    printf("%s: the oracle restatement %s liquid-dsp on this input\n", ok ? "PASS" : "FAIL", ok ? "matches" : "does NOT match");
    return ok ? 0 : 1;
}
