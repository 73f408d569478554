#!/bin/bash
# session-2 batch 3 (GPU box): what the front end's output stores cost (kernel alone, blocks not pipelined)
mkdir -p gpurun_out/s2
for W in cfg5 cfg2; do
  bash tools/variant_kstats.sh $W "-DPMR_BASELINE" "-DFE_OUT_AND=511ull" "-DFE_OUT_AND=0x3ffffull" "-DFE_OUT_NT" "-DPMR_BASELINE" 2>&1 | grep -E "^==|k_fe_fast|k_fe_stream|BUILD"
  PMR_FE_STREAM=8 bash tools/variant_kstats.sh $W "-DPMR_BASELINE" "-DFE_OUT_AND=511ull" "-DFE_STOP=3" 2>&1 | grep -E "^==|k_fe_fast|k_fe_stream|BUILD"
done > gpurun_out/s2/out_cost.txt 2>&1
