#!/bin/bash
# session-2 batch 6 (GPU box): 8192-sample front-end tiles -- parity, kernel alone, chain A/B
mkdir -p gpurun_out/s2
timeout 1200 python -m pytest tests/test_gpu_fe_tiles.py -x -q 2>&1 | tail -15 > gpurun_out/s2/nt512_test.txt
for W in cfg5 cfg3 cfg2; do
  for T in 256 512; do
    echo "== $W PMR_FE_NT=$T, blocks not pipelined"
    PMR_FE_NT=$T PMR_OVERLAP=0 bash tools/kstats.sh vk_tmp.txt --workload $W --also none --no-cpu-baseline --regions 2 --parity-blocks 0 --no-kernel-events
    head -6 gpurun_out/vk_tmp.txt
  done
done > gpurun_out/s2/nt512_kstats.txt 2>&1
for W in cfg5 cfg3 cfg2; do
  echo "#### $W"
  BENCH_ARGS="--regions 5" bash tools/env_ab.sh $W "PMR_FE_NT=256" "PMR_FE_NT=512" "PMR_FE_NT=256" "PMR_FE_NT=512"
done > gpurun_out/s2/nt512_ab.txt 2>&1
