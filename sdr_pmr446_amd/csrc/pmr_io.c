/* pmr_io.c -- recorded-IQ reader and WAV / raw PCM writer (include/pmr_io.h).
 * Host-only; stands in for the SoapySDR ingest (reference src/shared.c:11-88, readStream src/sdr_pmr446.c:789) and the
 * RtAudio / stdout sinks (src/sdr_pmr446.c:545-603, src/dsd_in.c:172-178) so the chain can run headless on recordings. */
#include "../../include/pmr_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct pmr_iq_reader_s { FILE *f; int own, format; unsigned char *raw; size_t raw_cap; };

static size_t iq_bytes(int format) { return format == PMR_IQ_CF32 ? 8 : format == PMR_IQ_CS16 ? 4 : 2; }

pmr_iq_reader pmr_iq_reader_open(const char *path, int format)
{
    if (!path || format < PMR_IQ_CF32 || format > PMR_IQ_CU8) return NULL;
    pmr_iq_reader r = (pmr_iq_reader)calloc(1, sizeof(*r));
    if (!r) return NULL;
    r->format = format;
    if (!strcmp(path, "-")) { r->f = stdin; r->own = 0; }
    else { r->f = fopen(path, "rb"); r->own = 1; }
    if (!r->f) { free(r); return NULL; }
    return r;
}

int pmr_iq_reader_read(pmr_iq_reader r, pmr_cf32 *buf, unsigned max_samples)
{
    if (!r || (!buf && max_samples)) return -1;
    if (max_samples > 0x7fffffffu) max_samples = 0x7fffffffu;
    const size_t bps = iq_bytes(r->format);
    float *out = (float *)buf;
    if (r->format == PMR_IQ_CF32) {
        /* fread may return short counts on pipes before end of stream: loop until full or EOF */
        size_t got = 0;
        while (got < max_samples) {
            size_t k = fread((char *)out + got * bps, bps, max_samples - got, r->f);
            if (k == 0) break;
            got += k;
        }
        if (got == 0 && ferror(r->f)) return -2;
        return (int)got;
    }
    if (r->raw_cap < (size_t)max_samples * bps) {
        unsigned char *nr = (unsigned char *)realloc(r->raw, (size_t)max_samples * bps);
        if (!nr) return -3;
        r->raw = nr; r->raw_cap = (size_t)max_samples * bps;
    }
    size_t got = 0;
    while (got < max_samples) {
        size_t k = fread(r->raw + got * bps, bps, max_samples - got, r->f);
        if (k == 0) break;
        got += k;
    }
    if (got == 0 && ferror(r->f)) return -2;
    if (r->format == PMR_IQ_CS16) {
        const int16_t *s = (const int16_t *)r->raw;
        for (size_t i = 0; i < 2 * got; i++) out[i] = (float)s[i] * (1.0f / 32768.0f);
    } else {
        for (size_t i = 0; i < 2 * got; i++) out[i] = ((float)r->raw[i] - 127.5f) * (1.0f / 127.5f);
    }
    return (int)got;
}

int pmr_iq_reader_close(pmr_iq_reader r)
{
    if (!r) return PMR_OK;
    if (r->own && r->f) fclose(r->f);
    free(r->raw); free(r);
    return PMR_OK;
}

/* ---- WAV / raw writer ---- */

struct pmr_wav_writer_s { FILE *f; int own, format; unsigned rate, channels; uint64_t data_bytes; void *tmp; size_t tmp_cap; };

static void put_u32(unsigned char *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = v >> 24; }
static void put_u16(unsigned char *p, unsigned v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

static int wav_header(pmr_wav_writer w)
{
    unsigned char h[44];
    const unsigned bytes = w->format == PMR_WAV_F32 ? 4 : 2;
    const uint64_t db = w->data_bytes > 0xffffffffull - 36 ? 0xffffffffull - 36 : w->data_bytes;
    memcpy(h, "RIFF", 4); put_u32(h + 4, (uint32_t)(36 + db)); memcpy(h + 8, "WAVEfmt ", 8);
    put_u32(h + 16, 16); put_u16(h + 20, w->format == PMR_WAV_F32 ? 3 : 1); put_u16(h + 22, w->channels);
    put_u32(h + 24, w->rate); put_u32(h + 28, w->rate * w->channels * bytes); put_u16(h + 32, w->channels * bytes);
    put_u16(h + 34, 8 * bytes); memcpy(h + 36, "data", 4); put_u32(h + 40, (uint32_t)db);
    return fwrite(h, 1, sizeof(h), w->f) == sizeof(h) ? PMR_OK : PMR_EINVAL;
}

pmr_wav_writer pmr_wav_writer_open(const char *path, int format, unsigned sample_rate, unsigned channels)
{
    if (!path || format < PMR_WAV_F32 || format > PMR_RAW_S16 || !sample_rate || !channels || channels > 65535) return NULL;
    const int to_stdout = !strcmp(path, "-");
    if (to_stdout && format != PMR_RAW_S16) return NULL;      /* a RIFF header needs a seekable file */
    pmr_wav_writer w = (pmr_wav_writer)calloc(1, sizeof(*w));
    if (!w) return NULL;
    w->format = format; w->rate = sample_rate; w->channels = channels;
    if (to_stdout) { w->f = stdout; w->own = 0; }
    else { w->f = fopen(path, "wb"); w->own = 1; }
    if (!w->f) { free(w); return NULL; }
    if (format != PMR_RAW_S16 && wav_header(w)) { if (w->own) fclose(w->f); free(w); return NULL; }
    return w;
}

static void *wav_tmp(pmr_wav_writer w, size_t bytes)
{
    if (w->tmp_cap < bytes) {
        void *n = realloc(w->tmp, bytes);
        if (!n) return NULL;
        w->tmp = n; w->tmp_cap = bytes;
    }
    return w->tmp;
}

int pmr_wav_writer_write_f32(pmr_wav_writer w, const float *data, unsigned frames, unsigned stride)
{
    if (!w || w->format != PMR_WAV_F32 || (!data && frames)) return PMR_EINVAL;
    if (!frames) return PMR_OK;
    const unsigned C = w->channels;
    const float *src = data;
    if (C > 1) {
        float *t = (float *)wav_tmp(w, (size_t)frames * C * sizeof(float));
        if (!t) return PMR_ENOMEM;
        for (unsigned c = 0; c < C; c++) for (unsigned i = 0; i < frames; i++) t[(size_t)i * C + c] = data[(size_t)c * stride + i];
        src = t;
    }
    if (fwrite(src, sizeof(float), (size_t)frames * C, w->f) != (size_t)frames * C) return PMR_EINVAL;
    w->data_bytes += (uint64_t)frames * C * sizeof(float);
    return PMR_OK;
}

int pmr_wav_writer_write_s16(pmr_wav_writer w, const int16_t *data, unsigned frames, unsigned stride)
{
    if (!w || w->format == PMR_WAV_F32 || (!data && frames)) return PMR_EINVAL;
    if (!frames) return PMR_OK;
    const unsigned C = w->channels;
    const int16_t *src = data;
    if (C > 1) {
        int16_t *t = (int16_t *)wav_tmp(w, (size_t)frames * C * sizeof(int16_t));
        if (!t) return PMR_ENOMEM;
        for (unsigned c = 0; c < C; c++) for (unsigned i = 0; i < frames; i++) t[(size_t)i * C + c] = data[(size_t)c * stride + i];
        src = t;
    }
    if (fwrite(src, sizeof(int16_t), (size_t)frames * C, w->f) != (size_t)frames * C) return PMR_EINVAL;
    w->data_bytes += (uint64_t)frames * C * sizeof(int16_t);
    if (!w->own) fflush(w->f);                                  /* dsd_in flushes every block, src/dsd_in.c:178 */
    return PMR_OK;
}

int pmr_wav_writer_close(pmr_wav_writer w)
{
    if (!w) return PMR_OK;
    int rc = PMR_OK;
    if (w->format != PMR_RAW_S16) {
        if (fseek(w->f, 0, SEEK_SET) != 0 || wav_header(w)) rc = PMR_EINVAL;
    }
    if (w->own) { if (fclose(w->f) != 0) rc = PMR_EINVAL; } else fflush(w->f);
    free(w->tmp); free(w);
    return rc;
}
