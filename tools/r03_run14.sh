mkdir -p gpurun_out
for W in cfg2 cfg3; do
  BENCH_ARGS="--no-kernel-events" bash tools/env_ab.sh $W "PMR_X=0" "PMR_TILEFIX_STREAM=fe" "PMR_X=0" "PMR_TILEFIX_STREAM=fe" > gpurun_out/r14_ab_$W.txt 2>&1
done
cat gpurun_out/r14_ab_*.txt
