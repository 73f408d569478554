#!/bin/bash
# usage (GPU box): bash tools/ref_point_timeline.sh : copies + kernels of synchronous 100000-sample calls (reference operating point)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/rpt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/ref_point_latency.py > $O/run.log 2>&1
python3 - $O > $R/gpurun_out/ref_point_timeline.txt <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
for f in glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?"))[:30]))
ev.sort()
# the pinned synchronous phase: calls 340..620 of the run are pinned sync calls; print a window from the middle of them
names = [e[2] for e in ev]
idx = [i for i, n in enumerate(names) if n.startswith("COPY")]
mid = idx[len(idx) // 4]       # inside the synchronous phases
t0 = ev[mid][0]
for s, e, n in ev[mid:mid + 30]:
    print("%9.1f %9.1f  %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
tail -5 $O/run.log >> $R/gpurun_out/ref_point_timeline.txt
rm -rf $O
