mkdir -p gpurun_out
for W in cfg3 cfg5 cfg2; do
  BENCH_ARGS="--no-kernel-events" bash tools/env_ab.sh $W "PMR_X=0" "PMR_STREAM_PRIO=fe" "PMR_X=0" "PMR_STREAM_PRIO=fe" > gpurun_out/r18_ab_$W.txt 2>&1
done
cat gpurun_out/r18_ab_cfg3.txt gpurun_out/r18_ab_cfg5.txt gpurun_out/r18_ab_cfg2.txt
