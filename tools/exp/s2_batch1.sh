#!/bin/bash
# session-2 batch 1 (GPU box): CU masks, any-order front-end launches, front end with / without its output phase
mkdir -p gpurun_out/s2
bash tools/exp/streams_ab.sh cfg5 cfg3 cfg2 > gpurun_out/s2/streams_ab.txt 2>&1
for W in cfg5 cfg3 cfg2; do
  PMR_CC_FLAGS= bash tools/variant_bench.sh $W "-DPMR_BASELINE" "-DFE_ANYORDER" "-DPMR_BASELINE" "-DFE_ANYORDER"
done > gpurun_out/s2/anyorder.txt 2>&1
PMR_HIPCC_FLAGS="-fno-slp-vectorize -DFE_ANYORDER" python3 sdr_pmr446_amd/build.py --force > /dev/null 2>&1
for W in cfg5 cfg2; do bash tools/fe_gaps.sh $W; done > gpurun_out/s2/anyorder_gaps.txt 2>&1
for W in cfg5 cfg2; do
  bash tools/variant_kstats.sh $W "-DPMR_BASELINE" "-DFE_STOP=3" "-DFE_STOP=1" "-DPMR_BASELINE"
done > gpurun_out/s2/fe_stop.txt 2>&1
