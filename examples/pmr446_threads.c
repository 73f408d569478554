/* pmr446_threads -- the deployment model include/pmr_chain.h promises, exercised from C: "one thread per handle; handles are
 * independent" (SURVEY.md s8(b) / (e)).  The reference is a threaded C program (audio callback thread src/sdr_pmr446.c:520-544
 * beside the DSP loop :788-931); a multi-receiver build of it runs one such loop per IQ stream, each on its own pthread with its
 * own handle.
 *
 *   pmr446_threads [n_threads = 2] [blocks = 12] [device = -1] [fs_in = 2.4e6] [num_channels = 16]
 *
 * Stream t (synthetic, seed t, generated in HBM by the library and copied to the host) is cut into ragged blocks and run twice
 * through a fresh handle: SERIALLY (thread 0 .. n-1 one after the other on the main thread) and CONCURRENTLY (n pthreads, all
 * started together behind a barrier, every handle on the same device).  Odd streams use the synchronous host call
 * (pmr_chain_process_block_f32), even streams the un-synchronised device call (pmr_chain_process_block_device) with pinned
 * staging of their own; stream 1 additionally runs the CTCSS detector and changes its channel mask and resets a channel between
 * blocks.  Every byte of PCM and every CTCSS event of the concurrent run must equal the serial run's.  Exit code 0 = identical.
 *
 * Everything numerical happens in libpmr446_hip.so; this file is plumbing. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pmr_chain.h"
#include "pmr_mem.h"

typedef struct {
    int id, device, nblocks;
    double fs; unsigned M;
    size_t n_total;
    pmr_cf32 *x;                 /* host copy of the stream (pmr_host_alloc) */
    unsigned *split;             /* block sizes */
    /* results */
    int16_t *pcm; size_t pcm_len, pcm_cap;          /* all blocks' PCM, [block][M][frames] flattened */
    pmr_ctcss_event *ev; size_t ev_len, ev_cap;
    int rc; char err[256];
    pthread_barrier_t *start;
} job;

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void fail_job(job *j, pmr_chain q, const char *what, int rc)
{
    j->rc = rc ? rc : -1;
    snprintf(j->err, sizeof(j->err), "stream %d: %s: %s", j->id, what, q ? pmr_chain_last_error(q) : pmr_chain_create_error());
}

static void *run_stream(void *arg)
{
    job *j = (job *)arg;
    j->rc = 0; j->pcm_len = 0; j->ev_len = 0;
    const unsigned M = j->M;
    unsigned mb = 0;
    for (int b = 0; b < j->nblocks; b++) if (j->split[b] > mb) mb = j->split[b];
    pmr_chain_cfg cfg;
    pmr_chain_default_cfg(&cfg);
    cfg.fs_in = j->fs; cfg.num_channels = M; cfg.max_block = mb; cfg.device = j->device;
    if (j->start) pthread_barrier_wait(j->start);              /* every thread creates its handle and streams at the same time */
    pmr_chain q = pmr_chain_create(&cfg);
    if (!q) { fail_job(j, NULL, "pmr_chain_create", -1); return NULL; }
    const unsigned S = pmr_chain_max_frames(q);
    const int with_ctcss = j->id == 1, device_calls = (j->id & 1) == 0;
    const unsigned ev_cap = S / 2441 + 2;
    int16_t *pcm = (int16_t *)calloc((size_t)M * S, sizeof(int16_t));
    pmr_ctcss_event *ev = (pmr_ctcss_event *)calloc((size_t)M * ev_cap, sizeof(*ev));
    void *d_iq = NULL, *d_pcm = NULL;
    int rc = 0;
    if (!pcm || !ev) { fail_job(j, q, "calloc", -1); goto out; }
    if (with_ctcss && (rc = pmr_chain_ctcss_enable(q, 1))) { fail_job(j, q, "ctcss_enable", rc); goto out; }
    if (device_calls) {
        d_iq = pmr_device_alloc((size_t)j->n_total * sizeof(pmr_cf32), j->device);
        d_pcm = pmr_device_alloc((size_t)M * S * sizeof(int16_t), j->device);
        if (!d_iq || !d_pcm || pmr_memcpy_h2d(d_iq, j->x, (size_t)j->n_total * sizeof(pmr_cf32))) { fail_job(j, q, "device buffers", -1); goto out; }
    }
    size_t pos = 0;
    for (int b = 0; b < j->nblocks; b++) {
        const unsigned n = j->split[b];
        unsigned ns = 0;
        if (with_ctcss) {
            /* the squelch's side of the interface between blocks: mask changes and a per-channel reset (:834-839, :866-867) */
            if (b % 4 == 1) { const uint64_t m = 0x0000000000000f0full; rc = pmr_chain_set_channel_mask(q, &m, 1); }
            else if (b % 4 == 3) rc = pmr_chain_set_channel_mask(q, NULL, 0);
            if (!rc && b % 3 == 2) rc = pmr_chain_reset_channel(q, (unsigned)b % M);
            if (rc) { fail_job(j, q, "mask / reset", rc); goto out; }
            memset(pcm, 0, (size_t)M * S * sizeof(int16_t));  /* rows of closed channels are left untouched by the library */
        }
        if (device_calls) {
            rc = pmr_chain_process_block_device(q, (const char *)d_iq + pos * sizeof(pmr_cf32), n, d_pcm, NULL, S, &ns, NULL, NULL);
            /* (two blocks in three stay un-synchronised: the two-stream pipeline of this handle runs beside the other threads') */
            if (!rc && (b % 3 == 2 || b == j->nblocks - 1)) rc = pmr_chain_synchronize(q);
        } else {
            rc = pmr_chain_process_block_f32(q, j->x + pos, n, pcm, NULL, S, &ns, NULL, NULL);
        }
        if (rc) { fail_job(j, q, "process_block", rc); goto out; }
        pos += n;
        if (device_calls) {
            /* PCM of the blocks that were synchronised (the buffer is overwritten by every call) */
            if (b % 3 == 2 || b == j->nblocks - 1) {
                if ((rc = pmr_memcpy_d2h(pcm, d_pcm, (size_t)M * S * sizeof(int16_t)))) { fail_job(j, q, "d2h", rc); goto out; }
            } else continue;
        }
        if (j->pcm_len + (size_t)M * ns > j->pcm_cap) { fail_job(j, q, "result capacity", -1); goto out; }
        for (unsigned k = 0; k < M; k++) memcpy(j->pcm + j->pcm_len + (size_t)k * ns, pcm + (size_t)k * S, (size_t)ns * sizeof(int16_t));
        j->pcm_len += (size_t)M * ns;
        if (with_ctcss) {
            unsigned nev = 0;
            if ((rc = pmr_chain_ctcss_read(q, ev, ev_cap, &nev))) { fail_job(j, q, "ctcss_read", rc); goto out; }
            for (unsigned k = 0; k < M; k++) for (unsigned e = 0; e < nev; e++) {
                if (j->ev_len >= j->ev_cap) { fail_job(j, q, "event capacity", -1); goto out; }
                j->ev[j->ev_len++] = ev[(size_t)k * ev_cap + e];
            }
        }
    }
out:
    if (d_iq) pmr_device_free(d_iq);
    if (d_pcm) pmr_device_free(d_pcm);
    free(pcm); free(ev);
    pmr_chain_destroy(q);
    return NULL;
}

int main(int argc, char **argv)
{
    const int nthr = argc > 1 ? atoi(argv[1]) : 2, nblocks = argc > 2 ? atoi(argv[2]) : 12, device = argc > 3 ? atoi(argv[3]) : -1;
    const double fs = argc > 4 ? atof(argv[4]) : 2.4e6;
    const unsigned M = argc > 5 ? (unsigned)atoi(argv[5]) : 16;
    if (nthr < 1 || nthr > 16 || nblocks < 1 || nblocks > 4096) { fprintf(stderr, "usage: pmr446_threads [threads 1..16] [blocks] [device] [fs_in] [M]\n"); return 2; }
    job *serial = (job *)calloc((size_t)nthr, sizeof(job)), *conc = (job *)calloc((size_t)nthr, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthr, sizeof(pthread_t));
    if (!serial || !conc || !th) return 3;
    const double rate = (double)M * 12500.0 / fs;
    for (int t = 0; t < nthr; t++) {
        job *j = &serial[t];
        j->id = t; j->device = device; j->nblocks = nblocks; j->fs = fs; j->M = M;
        j->split = (unsigned *)calloc((size_t)nblocks, sizeof(unsigned));
        if (!j->split) return 3;
        unsigned seed = 12345u + 77u * (unsigned)t;
        j->n_total = 0;
        for (int b = 0; b < nblocks; b++) {
            /* ragged: 0, 1, 7 samples, odd sizes, and blocks of a few hundred thousand samples */
            const unsigned r = lcg(&seed) % 16u;
            j->split[b] = r == 0 ? 0u : r == 1 ? 1u : r == 2 ? 7u : 60000u + lcg(&seed) % 400000u;
            j->n_total += j->split[b];
        }
        if (j->n_total == 0) { j->split[0] = 100000; j->n_total = 100000; }
        j->x = (pmr_cf32 *)pmr_host_alloc(j->n_total * sizeof(pmr_cf32));
        void *d = pmr_device_alloc(j->n_total * sizeof(pmr_cf32), device);
        pmr_synth_cfg sc;
        pmr_synth_default_cfg(&sc, fs, M);
        sc.stream_id = (unsigned)t; sc.dev_hz = 1500.0; sc.ctcss_dev_hz = 700.0;
        if (!j->x || !d || pmr_synth_iq_device(&sc, d, 0, j->n_total) || pmr_memcpy_d2h(j->x, d, j->n_total * sizeof(pmr_cf32))) {
            fprintf(stderr, "pmr446_threads: cannot synthesise stream %d (no HIP device?)\n", t); return 3;
        }
        pmr_device_free(d);
        j->pcm_cap = (size_t)((double)j->n_total * rate) + (size_t)M * (size_t)(nblocks + 4);
        j->ev_cap = (size_t)M * (size_t)((double)j->n_total * rate / (2441.0 * M) + nblocks + 4);
        j->pcm = (int16_t *)calloc(j->pcm_cap, sizeof(int16_t));
        j->ev = (pmr_ctcss_event *)calloc(j->ev_cap, sizeof(pmr_ctcss_event));
        conc[t] = *j;
        conc[t].pcm = (int16_t *)calloc(j->pcm_cap, sizeof(int16_t));
        conc[t].ev = (pmr_ctcss_event *)calloc(j->ev_cap, sizeof(pmr_ctcss_event));
        if (!j->pcm || !j->ev || !conc[t].pcm || !conc[t].ev) return 3;
    }
    for (int t = 0; t < nthr; t++) {                           /* serial reference: one stream after the other, this thread */
        run_stream(&serial[t]);
        if (serial[t].rc) { fprintf(stderr, "pmr446_threads (serial): %s\n", serial[t].err); return 4; }
    }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)nthr);
    for (int t = 0; t < nthr; t++) { conc[t].start = &bar; if (pthread_create(&th[t], NULL, run_stream, &conc[t])) return 3; }
    for (int t = 0; t < nthr; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    int bad = 0;
    for (int t = 0; t < nthr; t++) {
        if (conc[t].rc) { fprintf(stderr, "pmr446_threads (concurrent): %s\n", conc[t].err); bad = 1; continue; }
        const int same_pcm = conc[t].pcm_len == serial[t].pcm_len && !memcmp(conc[t].pcm, serial[t].pcm, serial[t].pcm_len * sizeof(int16_t));
        const int same_ev = conc[t].ev_len == serial[t].ev_len && !memcmp(conc[t].ev, serial[t].ev, serial[t].ev_len * sizeof(pmr_ctcss_event));
        long peak = 0, detected = 0;
        for (size_t i = 0; i < serial[t].pcm_len; i++) { const long a = labs((long)serial[t].pcm[i]); if (a > peak) peak = a; }
        for (size_t i = 0; i < serial[t].ev_len; i++) detected += serial[t].ev[i].detected;
        printf("stream %d: %zu samples in %d blocks (%s calls%s) -> %zu PCM samples (peak %ld), %zu CTCSS events (%ld detected): %s\n", t,
               serial[t].n_total, nblocks, (t & 1) ? "host" : "device", t == 1 ? ", CTCSS + mask changes + channel resets" : "",
               serial[t].pcm_len, peak, serial[t].ev_len, detected, same_pcm && same_ev ? "identical" : "DIFFERENT");
        if (!same_pcm || !same_ev || peak < 1000) bad = 1;
    }
    return bad ? 1 : 0;
}
