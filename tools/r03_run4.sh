mkdir -p gpurun_out
for W in cfg2 cfg3; do
  BENCH_ARGS="--no-kernel-events" bash tools/env_ab.sh $W "PMR_FIR_MFMA=32" "PMR_FIR_MFMA=32 PMR_STREAM_PRIO=1" "PMR_FIR_MFMA=tiles PMR_STREAM_PRIO=1" "PMR_STREAM_PRIO=1" "PMR_FIR_MFMA=32" "PMR_FIR_MFMA=32 PMR_STREAM_PRIO=1" "PMR_FIR_MFMA=tiles PMR_STREAM_PRIO=1" "PMR_STREAM_PRIO=1" > gpurun_out/r4_ab_$W.txt 2>&1
done
cat gpurun_out/r4_ab_*.txt
