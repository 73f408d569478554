#!/bin/bash
# session-2 batch 7 (GPU box): wave priority of the back-end kernels
mkdir -p gpurun_out/s2
for W in cfg5 cfg3 cfg2; do
  echo "#### $W"
  bash tools/variant_bench.sh $W "-DPMR_BASELINE" "-DBE_SETPRIO=1" "-DBE_SETPRIO=3" "-DPMR_BASELINE" "-DBE_SETPRIO=1" "-DBE_SETPRIO=3"
done > gpurun_out/s2/be_prio.txt 2>&1
