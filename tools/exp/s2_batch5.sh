#!/bin/bash
# session-2 batch 5 (GPU box): k_fe_loop -- parity, kernel alone, chain A/B per tiles-per-workgroup
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_fe_loop.py -x -q 2>&1 | tail -15 > gpurun_out/s2/loop_test.txt
for W in cfg5 cfg2; do
  for T in 1 2 4 8 16; do
    echo "== $W PMR_FE_TPW=$T, blocks not pipelined"
    PMR_FE_TPW=$T PMR_OVERLAP=0 bash tools/kstats.sh vk_tmp.txt --workload $W --also none --no-cpu-baseline --regions 2 --parity-blocks 0 --no-kernel-events
    grep -E "k_fe_fast|k_fe_loop" gpurun_out/vk_tmp.txt
  done
done > gpurun_out/s2/loop_kstats.txt 2>&1
for W in cfg5 cfg3 cfg2; do
  echo "#### $W"
  BENCH_ARGS="--regions 5" bash tools/env_ab.sh $W "PMR_FE_TPW=1" "PMR_FE_TPW=2" "PMR_FE_TPW=4" "PMR_FE_TPW=8" "PMR_FE_TPW=16" "PMR_FE_TPW=1" "PMR_FE_TPW=4"
done > gpurun_out/s2/loop_ab.txt 2>&1
