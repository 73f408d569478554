mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r2_test.txt
for W in cfg2 cfg3 cfg5; do
  BENCH_ARGS="" bash tools/env_ab.sh $W "PMR_X=0" "PMR_FIR_MFMA=32" "PMR_CARRY=inplace" "PMR_X=0" "PMR_FIR_MFMA=32" "PMR_CARRY=inplace" > gpurun_out/r2_ab_$W.txt 2>&1
  bash tools/quick_bench.sh $W > gpurun_out/r2_iso_$W.txt 2>&1
  PMR_FIR_MFMA=32 bash tools/quick_bench.sh $W > gpurun_out/r2_iso32_$W.txt 2>&1
done
cat gpurun_out/r2_test.txt gpurun_out/r2_ab_*.txt gpurun_out/r2_iso*.txt
