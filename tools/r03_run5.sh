mkdir -p gpurun_out
for W in cfg2 cfg3; do
  BENCH_ARGS="--no-kernel-events" bash tools/env_ab.sh $W "PMR_FIR_MFMA=32" "PMR_FIR_MFMA=tiles" "PMR_FIR_RUN=2" "PMR_FIR_RUN=3" "PMR_FIR_RUN=4" "PMR_FIR_RUN=6" "PMR_FIR_MFMA=32" "PMR_FIR_MFMA=tiles" "PMR_FIR_RUN=2" "PMR_FIR_RUN=3" "PMR_FIR_RUN=4" "PMR_FIR_RUN=6"> gpurun_out/r5_ab_$W.txt 2>&1
  for R in 2 4; do PMR_FIR_RUN=$R bash tools/quick_bench.sh $W > gpurun_out/r5_iso_${W}_$R.txt 2>&1; done
done
cat gpurun_out/r5_ab_*.txt gpurun_out/r5_iso_*
