"""ORACLE (test infrastructure only): Python restatement of the reference's channel select + squelch logic.

find_max_rssi_channel  -- /root/reference/src/sdr_pmr446.c:668-700
state machine          -- /root/reference/src/sdr_pmr446.c:828-874 (proc_scanning / proc_tuned, 5 dB hysteresis :859,
                          lock_mode_max re-targeting :848-857)
float32 arithmetic like the C code (rssi_avg accumulates in float).
"""
import numpy as np

SCANNING, TUNED = 0, 1


def find_max_rssi_channel(rssi_db, channel_mask):
    max_i, ch_en = -1, 0
    rssi_max = np.float32(0.0)
    rssi_avg = np.float32(0.0)
    for i, r in enumerate(np.asarray(rssi_db, dtype=np.float32)):
        if not (channel_mask >> i) & 1:                  # channel_mask: a Python int of any width (bit i = channel i)
            continue
        ch_en += 1
        rssi_avg = np.float32(rssi_avg + r)
        if max_i >= 0:
            if r > rssi_max:
                rssi_max, max_i = r, i
        else:
            rssi_max, max_i = r, i
    if max_i >= 0:
        return max_i, np.float32(rssi_max - np.float32(rssi_avg / np.float32(ch_en)))
    return -1, None


class Squelch:
    def __init__(self):
        self.state, self.active_chan, self.rssi = SCANNING, -1, np.float32(0.0)

    def update(self, rssi_db, channel_mask, squelch_level, lock_mode_max):
        max_ch, max_rssi = find_max_rssi_channel(rssi_db, channel_mask)
        if max_rssi is not None:
            self.rssi = max_rssi
        before = self.active_chan
        if self.state == SCANNING:
            if self.rssi > np.float32(squelch_level):
                self.active_chan, self.state = max_ch, TUNED
        else:
            if lock_mode_max and self.active_chan != max_ch:
                self.active_chan = max_ch
            if self.rssi < np.float32(squelch_level) - np.float32(5.0):
                self.active_chan, self.state = -1, SCANNING
        return self.active_chan != before
