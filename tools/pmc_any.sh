#!/bin/bash
# usage (GPU box): bash tools/pmc_any.sh <workload> <kernel substring> "<counters of pass 1>" ["<counters of pass 2>" ...] : per-launch
# averages of arbitrary PMC counters for the kernels whose name contains the substring, blocks not pipelined (PMR_OVERLAP=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=$1; K=$2; shift 2; O=$R/gpurun_out/pmcany; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PMR_OVERLAP=0
B="--workload $W --also none --regions 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-kernel-events --parity-blocks 0"
i=0
for C in "$@"; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -- python3 $R/bench.py $B > /dev/null 2>&1
  i=$((i+1))
done
PMC_KERNEL="$K" python3 - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/pmcany"
K=os.environ["PMC_KERNEL"]
acc={}
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","")
        if K not in n: continue
        acc.setdefault(n,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for n,d in acc.items():
    print(n)
    for c,v in sorted(d.items()):
        print("   %-28s %16.0f   (%d launches)" % (c, sum(v)/len(v), len(v)))
PY
rm -rf $O
