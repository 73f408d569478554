#!/bin/bash
# usage (GPU box): bash tools/fe_store_cost.sh [workloads...] : what the front-end tile kernel waits for -- the kernel ALONE (blocks not
# pipelined, rocprofv3 durations) built up to a phase boundary (-DFE_STOP), with its output stores redirected into one 4 KB window
# (-DFE_OUT_AND: no HBM write traffic), predicated off (-DFE_OUT_SKIP), non-temporal (-DFE_OUT_NT) or system-scope write-through
# (-DFE_OUT_SC).  All but the baseline give WRONG results; timing only.  DESIGN.md 4.1, profiles/r04_ab_log.txt r4s.
for W in ${@:-cfg5 cfg2}; do
  bash tools/variant_kstats.sh $W "-DPMR_BASELINE" "-DFE_STOP=1" "-DFE_STOP=3" "-DFE_OUT_SKIP" "-DFE_OUT_AND=511ull" "-DFE_OUT_NT" "-DFE_OUT_SC" "-DPMR_BASELINE" 2>&1 |
    grep -E "^==|k_fe_fast|BUILD"
done
