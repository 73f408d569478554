mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_carry.py -x -q 2>&1 | tail -15 > gpurun_out/r1_test.txt
for W in cfg2 cfg3; do
  BENCH_ARGS="" bash tools/env_ab.sh $W "PMR_X=0" "PMR_CARRY=inplace" "PMR_X=0" "PMR_CARRY=inplace" > gpurun_out/r1_ab_$W.txt 2>&1
  PMR_CARRY=inplace bash tools/quick_bench.sh $W > gpurun_out/r1_iso_inplace_$W.txt 2>&1
  bash tools/quick_bench.sh $W > gpurun_out/r1_iso_atload_$W.txt 2>&1
done
cat gpurun_out/r1_test.txt gpurun_out/r1_ab_*.txt gpurun_out/r1_iso_*.txt
