#!/bin/bash
# session-2 batch 2 (GPU box): streaming front end -- parity, then chain A/B per tiles-per-workgroup, then kernel alone
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_stream_fe.py -x -q 2>&1 | tail -15 > gpurun_out/s2/stream_test.txt
for W in cfg5 cfg3 cfg2; do
  echo "#### $W"
  BENCH_ARGS="--regions 5" bash tools/env_ab.sh $W "PMR_FE_STREAM=0" "PMR_FE_STREAM=4" "PMR_FE_STREAM=8" "PMR_FE_STREAM=16" "PMR_FE_STREAM=0" "PMR_FE_STREAM=8"
done > gpurun_out/s2/stream_ab.txt 2>&1
for W in cfg5 cfg2; do
  for T in 0 8; do
    echo "== $W PMR_FE_STREAM=$T, blocks not pipelined"
    PMR_FE_STREAM=$T PMR_OVERLAP=0 bash tools/kstats.sh vk_tmp.txt --workload $W --also none --no-cpu-baseline --regions 2 --parity-blocks 0 --no-kernel-events
    head -5 gpurun_out/vk_tmp.txt
  done
done > gpurun_out/s2/stream_kstats.txt 2>&1
