#!/bin/bash
# usage (GPU box): bash tools/ref_point_timeline_async.sh : HIP API calls, copies and kernels of the submit / collect loop at the reference point
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/rpta; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O -- python3 $R/tools/ref_point_latency.py --async-only > $O/run.log 2>&1
python3 - $O > $R/gpurun_out/ref_point_timeline_async.txt <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
for f in glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")[:30]))
for f in glob.glob(O + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host " + r["Function"]))
ev.sort()
syncs = [i for i, e in enumerate(ev) if e[2] == "host hipEventSynchronize"]
i0 = syncs[len(syncs) // 2]
t0 = ev[i0][0]
for s, e, n in ev[i0:i0 + 110]:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
tail -2 $O/run.log >> $R/gpurun_out/ref_point_timeline_async.txt
cp $O/run.log $R/gpurun_out/rpta_run.log; rm -rf $O
