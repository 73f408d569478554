/* pmr_squelch.c -- SURVEY.md s8 row f1: channel select + squelch hysteresis on the per-channel RSSI that
 * pmr_chain_process_block() returns (rssi_db, == average_power() of every chan_bufs row, src/sdr_pmr446.c:330-336).
 * Host-only C: this is the consumer of the GPU's RSSI reduction, mirroring find_max_rssi_channel()
 * (src/sdr_pmr446.c:668-700) and the proc_scanning / proc_tuned state machine (:828-874). */
#include "../../include/pmr_chain.h"

void pmr_squelch_init(pmr_squelch *s)
{
    s->state = PMR_SCANNING;      /* g_chain.state = proc_scanning, :145 */
    s->active_chan = -1;          /* :146 */
    s->rssi = 0.0f;
}

/* Channel selection of the reference's scan (src/sdr_pmr446.c:668-700): among the OPEN channels, the strongest one and how far it
 * stands above the mean of the open channels' dB values; ties go to the lowest channel, the mean is summed in channel order in
 * float32 (the reference's arithmetic, bit for bit: tests/test_squelch.py, tests/golden/rssi_ref.npz).  The mask has the layout of
 * pmr_chain_set_channel_mask -- bit (k & 63) of word (k >> 6) opens channel k: the reference's single uint64_t (:18, :293-295)
 * extended to any M; NULL = every channel open -- and is walked word by word, bit by bit (closed words cost one test).
 * Returns the channel, or -1 when no channel is open or the mask is shorter than M channels (*margin_db untouched then). */
int pmr_find_max_rssi_channel(const float *rssi_db, unsigned M, const uint64_t *mask_words, unsigned n_words, float *margin_db)
{
    if (mask_words && (uint64_t)n_words * 64 < M) return -1;
    int best = -1;
    unsigned n_open = 0;
    float best_db = 0.0f, sum_db = 0.0f;
    for (unsigned w = 0; w * 64u < M; w++) {
        uint64_t open = mask_words ? mask_words[w] : ~(uint64_t)0;
        const unsigned left = M - w * 64u;
        if (left < 64u) open &= (((uint64_t)1 << left) - 1u);           /* channels beyond M do not exist */
        while (open) {
            const unsigned k = w * 64u + (unsigned)__builtin_ctzll(open);
            open &= open - 1u;                                          /* lowest open channel first: channel order */
            const float db = rssi_db[k];
            sum_db += db;
            n_open++;
            if (best < 0 || db > best_db) { best = (int)k; best_db = db; }
        }
    }
    if (best >= 0) *margin_db = best_db - sum_db / (float)n_open;
    return best;
}

/* :828-874.  Returns 1 when the active channel changed (tuned, hopped or detuned), 0 otherwise. */
int pmr_squelch_update(pmr_squelch *s, const float *rssi_db, unsigned M, const uint64_t *mask_words, unsigned n_words,
                       float squelch_level, int lock_mode_max)
{
    float max_rssi = s->rssi;
    const int max_ch = pmr_find_max_rssi_channel(rssi_db, M, mask_words, n_words, &max_rssi);
    const int before = s->active_chan;
    s->rssi = max_rssi;
    if (s->state == PMR_SCANNING) {
        if (s->rssi > squelch_level) {                 /* :834 */
            s->active_chan = max_ch;
            s->state = PMR_TUNED;
        }
    } else {
        if (lock_mode_max && s->active_chan != max_ch) /* :848-857 */
            s->active_chan = max_ch;
        if (s->rssi < squelch_level - 5.0f) {          /* :859 */
            s->active_chan = -1;
            s->state = PMR_SCANNING;
        }
    }
    return s->active_chan != before;
}
