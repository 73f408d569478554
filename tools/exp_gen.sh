echo small; bash tools/quick_bench.sh cfg2
echo generic; PMR_CHANNELIZER=generic bash tools/quick_bench.sh cfg2
bash tools/quick_bench.sh cfg3; bash tools/quick_bench.sh cfg5
PMR_CHANNELIZER=generic python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -1
