#!/bin/bash
# usage (GPU box): bash tools/fe_gaps.sh <workload> : the idle gap between consecutive front-end kernels (end of n -> start of n+1) in
# the pipelined loop, from a rocprofv3 kernel trace -- what a kernel boundary on the front-end stream costs per step
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-cfg5}; O=/tmp/feg_$$; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $W --also none --regions 2 --steps 100 --warmup 3 --no-cpu-baseline --no-host-io --no-kernel-events --parity-blocks 0 --no-one-open > $O/bench.log 2>&1
python3 - $O $W <<'PY'
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
fe = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "k_fe_fast" in n:
        fe.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
fe.sort()
gaps = [(fe[i + 1][0] - fe[i][1]) / 1e3 for i in range(len(fe) - 1)]
gaps = sorted(g for g in gaps if g < 200)          # (region boundaries are longer)
dur = sorted((e - s) / 1e3 for s, e in fe)
print("%s: %d front-end kernels; duration median %.1f us; gap to the next one: median %.2f us, p10 %.2f, p90 %.2f" %
      (sys.argv[2], len(fe), dur[len(dur) // 2], gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[9 * len(gaps) // 10]))
PY
rm -rf $O
