#!/bin/bash
# usage (GPU box): bash tools/pmc_insts.sh <workload> [ENV=VAL ...] : instruction counters of every kernel, blocks not pipelined
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-cfg2}; shift; O=$R/gpurun_out/pmci; rm -rf $O; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
export PMR_OVERLAP=0
B="--workload $W --also none --regions 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-kernel-events --parity-blocks 0"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/a -- python3 $R/bench.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT --output-format csv -d $O/b -- python3 $R/bench.py $B > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/pmci"
acc={}
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","")
        if not any(k in n for k in ("k_fe_","k_frontend")): continue
        acc.setdefault(n,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for n,d in acc.items():
    print(n)
    for c,v in sorted(d.items()):
        print("   %-26s %14.0f" % (c, sum(v)/len(v)))
PY
rm -rf $O
