mkdir -p gpurun_out
( timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_carry.py tests/test_gpu_ctcss.py tests/test_gpu_mask.py -x -q 2>&1 | tail -15 ) > gpurun_out/r3_test.txt
for W in cfg2 cfg3; do
  BENCH_ARGS="" bash tools/env_ab.sh $W "PMR_X=0" "PMR_FIR_MFMA=32" "PMR_FIR_MFMA=tiles" "PMR_STREAM_PRIO=fe" "PMR_X=0" "PMR_FIR_MFMA=32" "PMR_STREAM_PRIO=fe" > gpurun_out/r3_ab_$W.txt 2>&1
  bash tools/quick_bench.sh $W > gpurun_out/r3_iso_$W.txt 2>&1
done
cat gpurun_out/r3_test.txt gpurun_out/r3_ab_*.txt gpurun_out/r3_iso_*.txt
