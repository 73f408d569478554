#!/usr/bin/env python3
"""Print the kernel timeline (start/end relative, us) of a few steady-state steps from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("k_frontend", "k_channelize", "k_fir", "k_fe_", "k_rssi", "k_ct_", "k_pfb", "k_fft")):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][:40]))
rows.sort()
# take a window in the middle
mid = len(rows) // 2
win = rows[mid:mid + 24]
t0 = win[0][0]
for s, e, n in win:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
