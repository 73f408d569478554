#!/bin/bash
# usage (GPU box): bash tools/ref_point_timeline.sh : HIP API calls + kernels of synchronous 100000-sample calls (reference operating point)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/rpt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O -- python3 $R/tools/ref_point_latency.py --sync-only > $O/run.log 2>&1
python3 - $O > $R/gpurun_out/ref_point_timeline.txt <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
for f in glob.glob(O + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host " + r["Function"]))
ev.sort()
# a window of three calls from the last (pinned, synchronous) phase
syncs = [i for i, e in enumerate(ev) if e[2] in ("host hipEventSynchronize", "host hipStreamSynchronize")]
i0 = syncs[-6] + 1
t0 = ev[i0][0]
for s, e, n in ev[i0:syncs[-3] + 1]:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
tail -3 $O/run.log >> $R/gpurun_out/ref_point_timeline.txt
rm -rf $O
