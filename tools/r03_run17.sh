mkdir -p gpurun_out
( timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r17_test.txt
for W in cfg2 cfg3 cfg5; do bash tools/quick_bench.sh $W >> gpurun_out/r17_test.txt 2>&1; done
python3 tools/ref_point_latency.py >> gpurun_out/r17_test.txt 2>&1
cat gpurun_out/r17_test.txt
