#!/bin/bash
# session-2 batch 4 (GPU box): output-store policies of the front end (kernel alone, blocks not pipelined), then in the chain
mkdir -p gpurun_out/s2
for W in cfg5 cfg2; do
  bash tools/variant_kstats.sh $W "-DPMR_BASELINE" "-DFE_OUT_SKIP" "-DFE_OUT_NT" "-DFE_OUT_SC" "-DPMR_BASELINE" 2>&1 | grep -E "^==|k_fe_fast|k_fe_stream|BUILD"
done > gpurun_out/s2/out_policy.txt 2>&1
for W in cfg5 cfg2 cfg3; do
  bash tools/variant_bench.sh $W "-DPMR_BASELINE" "-DFE_OUT_NT" "-DFE_OUT_SC" "-DPMR_BASELINE" "-DFE_OUT_NT" "-DFE_OUT_SC"
done > gpurun_out/s2/out_policy_chain.txt 2>&1
