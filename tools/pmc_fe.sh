#!/bin/bash
# PMC counters of the front-end kernel, blocks not pipelined (clean attribution).  Run under gpurun.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmcfe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PMR_OVERLAP=0
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-kernel-events > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/pmcfe"
acc={}
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","")
        if not any(k in n for k in ("k_frontend","k_channelize","k_fir","k_fe_")): continue
        acc.setdefault(n,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for n,d in acc.items():
    print(n)
    for c,v in sorted(d.items()):
        print("   %-26s %14.0f" % (c, sum(v)/len(v)))
PY
