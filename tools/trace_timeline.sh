#!/bin/bash
# usage (GPU box): bash tools/trace_timeline.sh <workload> -> gpurun_out/tl_<workload>.txt : kernel timeline of steady-state steps + per-kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-cfg5}; O=$R/gpurun_out/tl_$W; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload $W --also none --regions 2 --steps 40 --warmup 3 --no-cpu-baseline --no-host-io --no-kernel-events --parity-blocks 0 > $O/bench.log 2>&1
python3 $R/tools/timeline.py $O > $R/gpurun_out/tl_$W.txt 2>&1
python3 - $O >> $R/gpurun_out/tl_$W.txt <<'PY'
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
agg = {}
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in ("k_frontend", "k_fe_", "k_channelize", "k_fir", "k_rssi", "k_ct_")):
        agg.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-40s calls %4d avg %8.1f us min %8.1f max %8.1f" % (n[:40], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
PY
grep '^{' $O/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'])" >> $R/gpurun_out/tl_$W.txt
rm -rf $O
