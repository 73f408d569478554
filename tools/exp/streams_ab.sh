#!/bin/bash
# usage (GPU box): bash tools/exp/streams_ab.sh [workloads...] : CU-mask / priority experiment of the two streams (library built with
# -DPMR_EXP_STREAMS: pmr_chain.c chain_create reads PMR_X_* from the environment)
PMR_CC_FLAGS=-DPMR_EXP_STREAMS python3 sdr_pmr446_amd/build.py --force > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; exit 1; }
./tools/exp/cumask_probe
for W in ${@:-cfg5 cfg3 cfg2}; do
  echo "#### $W"
  BENCH_ARGS="--regions 5" bash tools/env_ab.sh $W "PMR_X_NONE=1" \
    "PMR_X_BE_CUS=4" "PMR_X_BE_CUS=8" "PMR_X_BE_CUS=12" "PMR_X_BE_CUS=16" "PMR_X_BE_CUS=24" \
    "PMR_X_BE_CUS=8 PMR_X_FE_COMPL=1" "PMR_X_BE_CUS=4 PMR_X_FE_COMPL=1" \
    "PMR_X_BE_CUS=8 PMR_X_FE_PRIO=lo" "PMR_X_BE_CUS=16 PMR_X_FE_PRIO=lo" "PMR_X_BE_CUS=8 PMR_X_FE_PRIO=hi" "PMR_X_BE_CUS=16 PMR_X_FE_PRIO=hi" \
    "PMR_X_NONE=1" 2>&1 | grep -v "^PMR_EXP"
done
