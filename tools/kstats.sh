#!/bin/bash
# usage (GPU box): bash tools/kstats.sh <out.txt> <bench.py arguments...> : per-kernel calls / avg / min / max (us) of one bench.py run (rocprofv3 --kernel-trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; shift
O=/tmp/kstats_$$; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py "$@" > $O/run.log 2>&1
python3 - $O > $OUT <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-62s calls %6d  avg %8.1f  min %8.1f  med %8.1f  max %8.1f us" % (k, len(v), sum(v) / len(v), v[0], v[len(v) // 2], v[-1]))
PY
rm -rf $O
