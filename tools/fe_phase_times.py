#!/usr/bin/env python3
"""Where a front-end tile's time goes (diagnostic).  Needs a library built with -DFE_STAMP:
    python3 sdr_pmr446_amd/build.py --variant stamp "-DFE_STAMP"          (-> build_ab/stamp/, an experiment build)
    PMR_LIBRARY=build_ab/stamp/libpmr446_hip.so python3 tools/fe_phase_times.py cfg3
k_fe_fast then writes s_memtime (shader-clock cycles) at its phase boundaries for every tile; the last block's stamps are read back:
start -> tile landed (DMA) -> dc scan + first stage -> cascade -> resampler / stores, in cycles and as shares of a tile's lifetime.  Blocks run one at a time with the
handle's overlap off, so the front end is alone on the chip while it runs (the back end follows it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdr_pmr446_amd import chain
import bench
fs, M, lb, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
n = 1 << 26
g = chain.PmrChain(fs_in=fs, num_channels=M, max_block=n)
g.set_overlap(False)
dx = chain.synth_iq_device(n, fs, M)
stride = g.max_frames
dp = chain.DeviceBuffer(M * stride * 2)
for _ in range(3):
    g.process_block_device(dx.ptr, n, d_pcm=dp.ptr, stride=stride)
    chain.device_synchronize()
L = g._L
L.pmr_debug_fe_stamps.argtypes = [C.c_void_p, C.c_size_t]
nt = min(65536, (n + 3000) // 3600)
st = np.zeros((65536, 8), np.uint64)
assert L.pmr_debug_fe_stamps(st.ctypes.data, st.nbytes) == 0
st = st[: nt - 8].astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
ph = np.diff(st[:, :5], axis=1).astype(np.float64)          # cycles
names = ["wait for the tile (DMA)", "dc scan + first stage", "cascade", "resampler + stores"]
life = (st[:, 4] - st[:, 0]).astype(np.float64)
print("%d tiles" % len(st))
for i, nm in enumerate(names):
    print("  %-26s mean %7.0f  median %7.0f  p90 %7.0f cycles  (%4.1f %% of the lifetime)" % (nm, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), 100 * ph[:, i].mean() / life.mean()))
print("  %-26s mean %7.0f  median %7.0f  p90 %7.0f cycles" % ("tile lifetime", life.mean(), np.median(life), np.percentile(life, 90)))

# ---- slot turn-over: which CU ran each tile (XCC_ID / HW_ID stamped at the start), how many tiles a CU holds over time, and how long a
# slot stays empty between a tile's last stamp (stores issued) and the start of the next tile on that CU
hw = st[:, 5]
cu_key = ((hw >> 32) & 15) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)       # xcc, se, sh, cu
turn, alive_frac = [], []
for k in np.unique(cu_key):
    m = cu_key == k
    s0, e0 = np.sort(st[m, 0]), np.sort(st[m, 4])
    if len(s0) < 12:
        continue
    # the i-th end frees a slot that the (i + 4)-th start takes (four tiles per CU): greedy pairing in time order
    for i in range(len(e0) - 4):
        turn.append(s0[i + 4] - e0[i])
    span = e0[-1] - s0[0]
    alive_frac.append((st[m, 4] - st[m, 0]).sum() / max(1, span))
turn = np.array(turn, np.float64)
print("  %d CUs; tiles alive per CU (stamped lifetimes / span): mean %.2f" % (len(alive_frac), np.mean(alive_frac)))
print("  slot turn-over (a tile's last stamp -> start of the tile that takes its slot): mean %.0f  median %.0f  p90 %.0f cycles = %.1f %% of a lifetime"
      % (turn.mean(), np.median(turn), np.percentile(turn, 90), 100 * turn.mean() / life.mean()))
