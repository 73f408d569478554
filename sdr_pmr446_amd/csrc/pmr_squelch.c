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

/* :668-700 -- only mask-enabled channels take part; result is (max - mean of the dB values).  The mask has the layout of
 * pmr_chain_set_channel_mask (bit k & 63 of word k >> 6 enables channel k: the reference's uint64_t channel_mask, :18, :293-295,
 * for any M; NULL = every channel enabled).  A mask shorter than M channels is an error: -1, no channel. */
int pmr_find_max_rssi_channel(const float *rssi_db, unsigned M, const uint64_t *mask_words, unsigned n_words, float *max_rssi)
{
    int max_i = -1, ch_en = 0;
    float rssi_max = 0.0f, rssi_avg = 0.0f;
    if (mask_words && (uint64_t)n_words * 64 < M) return -1;
    for (unsigned i = 0; i < M; i++) {
        const int enabled = mask_words ? (int)((mask_words[i >> 6] >> (i & 63)) & 1u) : 1;
        if (!enabled) continue;
        ++ch_en;
        const float rssi = rssi_db[i];
        rssi_avg += rssi;
        if (max_i >= 0) {
            if (rssi > rssi_max) { rssi_max = rssi; max_i = (int)i; }
        } else {
            rssi_max = rssi; max_i = (int)i;
        }
    }
    if (max_i >= 0) {
        rssi_avg /= (float)ch_en;
        *max_rssi = rssi_max - rssi_avg;
    }
    return max_i;
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
